set -x
python bench.py --steps 30 --warmup 5 > gpurun_out/r02_bench_cfgB.json 2> gpurun_out/r02_bench_cfgB.err
bash tools/prof_bench.sh r02a > gpurun_out/r02a.txt 2>&1
bash tools/prof_bench.sh r02f --precision fp32 --steps 10 > gpurun_out/r02f.txt 2>&1
bash tools/prof_bench.sh r02t --mode train --precision bf16 --steps 10 --warmup 3 > gpurun_out/r02t.txt 2>&1
bash tools/prof_vit.sh r02v > gpurun_out/r02v.txt 2>&1
bash tools/pmc_traffic.sh gpurun_out/r02_traffic > gpurun_out/r02_traffic.txt 2>&1
python tools/bench_vit.py > gpurun_out/r02_vit.txt 2>&1
python bench.py --mode train --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r02_bench_cfgB_train.json 2>/dev/null
python bench.py --workload cfgA --no-cpu-baseline --steps 50 > gpurun_out/r02_bench_cfgA.json 2>/dev/null
python bench.py --workload cfgC --no-cpu-baseline --steps 20 > gpurun_out/r02_bench_cfgC.json 2>/dev/null
python bench.py --workload cam16 --no-cpu-baseline --headline-only --steps 50 > gpurun_out/r02_bench_cam16.json 2>/dev/null
python tools/gemm_bench.py cfgB cfgA vit > gpurun_out/r02_gemm_bench.txt 2>&1
tail -3 gpurun_out/r02_traffic.txt
