set -e
mkdir -p gpurun_out/prof
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1
tail -2 gpurun_out/t.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b.json 2> gpurun_out/b.err
grep "timed region" gpurun_out/b.err
for B in 16 8 4; do timeout -k 10 200 python bench.py --child --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/b_$B.json 2> gpurun_out/b_$B.err; grep "timed region" gpurun_out/b_$B.err; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r02 -- python3 $GRAFT_REPO_ROOT/bench.py --child --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --secondary --no-parity > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_step.py gpurun_out/prof 70 > gpurun_out/last_step_by_kernel.txt
python tools/trace_step.py gpurun_out/prof 80 grid > gpurun_out/last_step_by_kernel_and_grid.txt
head -3 gpurun_out/last_step_by_kernel.txt
rm -f gpurun_out/prof/*kernel_trace.csv gpurun_out/prof/*/*kernel_trace.csv
ls -la gpurun_out/prof gpurun_out/prof/* | head
