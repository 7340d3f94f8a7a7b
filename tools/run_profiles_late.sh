# Late round-6 refresh (GPU box, repo root): the measurements that changed after the main battery (tools/run_profiles.sh) -- the training
# step (TN kernel, hl chain), the ViT block (128-wide residual tiles), the default bench line.   bash tools/run_profiles_late.sh
R=r06; O=gpurun_out/${R}_late; mkdir -p $O
T="timeout 600"
$T python bench.py > $O/bench_cfgB.json 2> $O/bench_cfgB.err
$T bash tools/prof_bench.sh ${R}_train_f32 --mode train --precision fp32 --steps 20 --warmup 5 > $O/prof_train_f32.txt 2>&1
$T bash tools/prof_bench.sh ${R}_train_bf16 --mode train --precision bf16 --steps 20 --warmup 5 > $O/prof_train_bf16.txt 2>&1
$T bash tools/prof_bench.sh ${R}_vit_bf16 --workload vit --precision bf16 --steps 5 > $O/prof_vit_bf16.txt 2>&1
$T python bench.py --mode train --precision fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_cfgB_train_f32.json 2>/dev/null
$T python bench.py --mode train --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_cfgB_train_bf16.json 2>/dev/null
$T python bench.py --workload vit --steps 10 > $O/bench_vit.json 2>/dev/null
$T python bench.py --workload cfgA --no-cpu-baseline --steps 200 > $O/bench_cfgA.json 2>/dev/null
$T python tools/tn_chunks_bench.py > $O/gemm_tn.txt 2>&1
$T python tools/train_cpu_time.py fp32 >> $O/gemm_tn.txt 2>&1
$T python tools/train_cpu_time.py bf16 >> $O/gemm_tn.txt 2>&1
for n in train_f32 train_bf16 vit_bf16; do cp gpurun_out/prof_${R}_$n/p_kernel_stats.csv $O/${n}_kernel_stats.csv 2>/dev/null; done
head -c 900 $O/bench_cfgB.json; echo; head -c 300 $O/bench_cfgB_train_f32.json; echo; head -c 300 $O/bench_vit.json; echo; tail -14 $O/gemm_tn.txt
