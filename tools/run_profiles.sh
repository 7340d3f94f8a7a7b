# End-of-round measurement run (GPU box, repo root): bench lines, rocprofv3 kernel stats, PMC traffic of the attention kernels,
# micro-benchmarks.  Everything lands under gpurun_out/r06/; copy what is to be judged into profiles/.   bash tools/run_profiles.sh
R=r06; O=gpurun_out/$R; mkdir -p $O
T="timeout 600"
$T python bench.py > $O/bench_cfgB.json 2> $O/bench_cfgB.err
$T bash tools/prof_bench.sh ${R}_f32 --precision fp32 --steps 200 > $O/prof_f32.txt 2>&1
$T bash tools/prof_bench.sh ${R}_bf16 --precision bf16 --steps 200 > $O/prof_bf16.txt 2>&1
$T bash tools/prof_bench.sh ${R}_train_bf16 --mode train --precision bf16 --steps 20 --warmup 5 > $O/prof_train_bf16.txt 2>&1
$T bash tools/prof_bench.sh ${R}_train_f32 --mode train --precision fp32 --steps 20 --warmup 5 > $O/prof_train_f32.txt 2>&1
$T bash tools/prof_bench.sh ${R}_vit_bf16 --workload vit --precision bf16 --steps 5 > $O/prof_vit_bf16.txt 2>&1
$T bash tools/prof_bench.sh ${R}_vit_f32 --workload vit --precision fp32 --steps 3 > $O/prof_vit_f32.txt 2>&1
$T bash tools/pmc_traffic.sh $O/traffic > $O/traffic.txt 2>&1
$T bash tools/pmc_traffic_x3p.sh cfgB $O/traffic_x3p_cfgB > $O/traffic_x3p_cfgB.txt 2>&1
$T bash tools/pmc_traffic_x3p.sh cfgC $O/traffic_x3p_cfgC > $O/traffic_x3p_cfgC.txt 2>&1
$T bash tools/pmc_traffic_x3p.sh cfgA $O/traffic_x3p_cfgA > $O/traffic_x3p_cfgA.txt 2>&1
$T bash tools/pmc_x3p.sh $O/pmc_x3p > $O/attn_x3p_pmc_sq.txt 2>&1
WHAT=attnB $T bash tools/pmc_attn.sh $O/pmc_bf16 > /dev/null 2>&1; python tools/pmc_summary.py $O/pmc_bf16 sparse_attn_mfma > $O/attn_mfma_pmc_sq.txt 2>&1
$T python bench.py --mode train --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_cfgB_train_bf16.json 2>/dev/null
$T python bench.py --mode train --precision fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_cfgB_train_f32.json 2>/dev/null
$T python bench.py --workload cfgA --no-cpu-baseline --steps 200 > $O/bench_cfgA.json 2>/dev/null
$T python bench.py --workload cfgC --no-cpu-baseline --steps 50 > $O/bench_cfgC.json 2>/dev/null
$T python bench.py --workload cam16 --no-cpu-baseline --headline-only --steps 100 > $O/bench_cam16.json 2>/dev/null
$T python bench.py --workload vit --steps 10 > $O/bench_vit.json 2>/dev/null
for w in readme_dino_scratch readme_dino_adapter readme_mae_adapter; do $T python bench.py --workload $w --steps 200 --warmup 20 > $O/bench_$w.json 2>/dev/null; done
$T bash tools/prof_bench.sh ${R}_cfgA_f32 --workload cfgA --precision fp32 --steps 300 > $O/prof_cfgA_f32.txt 2>&1
$T bash tools/prof_bench.sh ${R}_cfgC_f32 --workload cfgC --precision fp32 --steps 50 > $O/prof_cfgC_f32.txt 2>&1
$T bash tools/prof_bench.sh ${R}_readme_mae_f32 --workload readme_mae_adapter --precision fp32 --steps 100 > $O/prof_readme_mae_f32.txt 2>&1
$T bash tools/prof_bench.sh ${R}_readme_scratch_f32 --workload readme_dino_scratch --precision fp32 --steps 100 > $O/prof_readme_scratch_f32.txt 2>&1
$T python tools/exact_attn_bench.py > $O/exact_attn_bench.txt 2>&1
$T python tools/exact_attn_bwd_bench.py >> $O/exact_attn_bench.txt 2>&1
$T python tools/gemm_bench.py cfgB cfgA vit > $O/gemm_bench.txt 2>&1
$T python tools/gemm_x3_bench.py cfgB cfgA vit > $O/gemm_x3_bench.txt 2>&1
$T python tools/topk_bench.py > $O/topk_bench.txt 2>&1
$T python tools/kbench.py x3 > $O/attn_x3_timing.txt 2>&1
$T python tools/varlen_bench.py > $O/varlen_bench.md 2> $O/varlen_bench.err
$T bash tools/prof_cmd.sh ${R}_varlen_1k_bf16 python tools/varlen_one.py 1000 384 bf16 64 > $O/prof_varlen_1k_bf16.txt 2>&1
$T bash tools/prof_cmd.sh ${R}_varlen_8k_f32 python tools/varlen_one.py 8192 384 fp32 16 > $O/prof_varlen_8k_f32.txt 2>&1
$T bash tools/pmc_vit.sh $O/pmc_vit > $O/vit_mfma_pmc.txt 2>&1
$T python tools/gemm_hl_splitk_time.py > $O/gemm_hl_splitk.txt 2>&1
for k in 128 200 256; do $T python tools/x3p_dev.py 32768 $k 6 --time 2>&1 | grep "x3"; done > $O/attn_x3p_timing.txt
$T python tools/x3p_dev.py 100000 512 6 --time 2>&1 | grep "x3" >> $O/attn_x3p_timing.txt
for sh in "8192 200 6 64" "32768 200 6 64" "100000 200 6 64" "1000 200 6 64"; do $T python tools/x3p_dev.py $sh --time 2>&1 | grep "x3"; done >> $O/attn_x3p_timing.txt
for n in f32 bf16 train_bf16 train_f32 vit_bf16 vit_f32 varlen_1k_bf16 varlen_8k_f32 cfgA_f32 cfgC_f32 readme_mae_f32 readme_scratch_f32; do cp gpurun_out/prof_${R}_$n/p_kernel_stats.csv $O/${n}_kernel_stats.csv 2>/dev/null; done
$T python tools/tn_chunks_bench.py > $O/gemm_tn.txt 2>&1
$T python tools/train_cpu_time.py fp32 >> $O/gemm_tn.txt 2>&1
$T python tools/train_cpu_time.py bf16 >> $O/gemm_tn.txt 2>&1
$T python tools/gemm_hl_l2_probe.py > $O/gemm_hl_l2_probe.txt 2>&1
timeout 1500 python tools/sweep.py > $O/sweep.md 2> $O/sweep.err
python -m pytest tests/test_gpu_vit.py tests/test_gpu_model.py -q -m gpu -s 2>&1 | grep MEASURED > $O/measured_errors.txt
tail -4 $O/traffic.txt; head -c 600 $O/bench_cfgB.json
