set -e
mkdir -p gpurun_out/prof4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -o r02b4 -- python3 $GRAFT_REPO_ROOT/bench.py --child --batch 4 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --secondary --no-parity > $GRAFT_REPO_ROOT/gpurun_out/prof4.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_step.py gpurun_out/prof4 90 > gpurun_out/last_step_b4.txt
TRACE_NO_MERGE=1 python tools/trace_step.py gpurun_out/prof4 120 > gpurun_out/last_step_b4_all.txt
head -3 gpurun_out/last_step_b4.txt
rm -f gpurun_out/prof4/*kernel_trace.csv gpurun_out/prof4/*/*kernel_trace.csv
