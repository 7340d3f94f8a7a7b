R=r06; O=gpurun_out/${R}_late2; mkdir -p $O
T="timeout 600"
$T bash tools/prof_bench.sh ${R}_cfgA_f32 --workload cfgA --precision fp32 --steps 300 > $O/prof_cfgA_f32.txt 2>&1
$T bash tools/prof_bench.sh ${R}_vit_f32 --workload vit --precision fp32 --steps 3 > $O/prof_vit_f32.txt 2>&1
$T bash tools/prof_bench.sh ${R}_vit_bf16 --workload vit --precision bf16 --steps 5 > $O/prof_vit_bf16.txt 2>&1
$T bash tools/prof_bench.sh ${R}_readme_mae_f32 --workload readme_mae_adapter --precision fp32 --steps 100 > $O/prof_readme_mae_f32.txt 2>&1
$T bash tools/prof_bench.sh ${R}_readme_scratch_f32 --workload readme_dino_scratch --precision fp32 --steps 100 > $O/prof_readme_scratch_f32.txt 2>&1
$T bash tools/prof_bench.sh ${R}_train_bf16 --mode train --precision bf16 --steps 20 --warmup 5 > $O/prof_train_bf16.txt 2>&1
$T bash tools/prof_bench.sh ${R}_train_f32 --mode train --precision fp32 --steps 20 --warmup 5 > $O/prof_train_f32.txt 2>&1
$T python bench.py --workload cfgA --no-cpu-baseline --steps 200 > $O/bench_cfgA.json 2>/dev/null
$T python bench.py --workload vit --steps 10 > $O/bench_vit.json 2>/dev/null
$T python bench.py --mode train --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_cfgB_train_bf16.json 2>/dev/null
$T python bench.py --mode train --precision fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_cfgB_train_f32.json 2>/dev/null
for w in readme_dino_scratch readme_dino_adapter readme_mae_adapter; do $T python bench.py --workload $w --steps 200 --warmup 20 > $O/bench_$w.json 2>/dev/null; done
$T python tools/gemm_bench.py cfgB cfgA vit > $O/gemm_bench.txt 2>&1
$T python tools/gemm_x3_bench.py cfgB cfgA vit > $O/gemm_x3_bench.txt 2>&1
for n in cfgA_f32 vit_f32 vit_bf16 readme_mae_f32 readme_scratch_f32 train_bf16 train_f32; do cp gpurun_out/prof_${R}_$n/p_kernel_stats.csv $O/${n}_kernel_stats.csv 2>/dev/null; done
for f in bench_cfgA bench_vit bench_cfgB_train_bf16 bench_cfgB_train_f32 bench_readme_dino_scratch bench_readme_dino_adapter bench_readme_mae_adapter; do head -c 120 $O/$f.json; echo; done
