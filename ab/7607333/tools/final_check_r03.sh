# round-3 measurement script (run on the GPU box from the repo root): driver-flag bench, local-batch sweep, rocprofv3
# kernel trace of the bench child at B = 32 and B = 4, two --pmc passes for the dominant kernel template
set -e
mkdir -p gpurun_out/prof gpurun_out/prof4 gpurun_out/pmc_f gpurun_out/pmc_w
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b.json 2> gpurun_out/b.err
grep "timed region" gpurun_out/b.err
cp gpurun_out/bench_kernel_table.json gpurun_out/b_kernel_table.json
for B in 16 8 4; do timeout -k 10 200 python bench.py --child --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/b_$B.json 2> gpurun_out/b_$B.err; grep "timed region" gpurun_out/b_$B.err; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r03 -- python3 $GRAFT_REPO_ROOT/bench.py --child --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --secondary --no-parity > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -o r03b4 -- python3 $GRAFT_REPO_ROOT/bench.py --child --batch 4 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --secondary --no-parity > $GRAFT_REPO_ROOT/gpurun_out/prof4.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_step.py gpurun_out/prof 70 > gpurun_out/last_step_by_kernel.txt
python tools/trace_step.py gpurun_out/prof 60 template > gpurun_out/last_step_by_template.txt
python tools/trace_step.py gpurun_out/prof 80 grid > gpurun_out/last_step_by_kernel_and_grid.txt
python tools/trace_step.py gpurun_out/prof4 75 > gpurun_out/last_step_b4.txt
head -3 gpurun_out/last_step_by_kernel.txt; head -3 gpurun_out/last_step_b4.txt
rm -f gpurun_out/prof/*kernel_trace.csv gpurun_out/prof/*/*kernel_trace.csv gpurun_out/prof4/*kernel_trace.csv gpurun_out/prof4/*/*kernel_trace.csv
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_f -o f -- python3 $GRAFT_REPO_ROOT/bench.py --child --steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --secondary --no-parity > $GRAFT_REPO_ROOT/gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_w -o w -- python3 $GRAFT_REPO_ROOT/bench.py --child --steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --secondary --no-parity > $GRAFT_REPO_ROOT/gpurun_out/pmc_w.log 2>&1
cd $GRAFT_REPO_ROOT
DOM=$(python -c "import json; print(json.loads(open('gpurun_out/b.json').read().strip().splitlines()[-1])['roofline']['kernel'])")
echo "dominant kernel template: $DOM"
python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w "$DOM" gpurun_out/pmc_dominant_traffic.json
rm -f gpurun_out/pmc_f/*/*counter_collection.csv gpurun_out/pmc_w/*/*counter_collection.csv
