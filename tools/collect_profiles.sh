#!/bin/bash
# Copy the judged artefacts of an end-of-round run (tools/run_profiles.sh + tools/sweep.py, merged back under gpurun_out/r06/) into
# profiles/ under their per-round names.  usage (container, repo root): bash tools/collect_profiles.sh
R=r06; O=gpurun_out/$R; P=profiles
cp_if() { [ -s "$1" ] && cp "$1" "$2"; }
cp_if $O/bench_cfgB.json $P/${R}_bench_cfgB.json
cp_if $O/bench_cfgA.json $P/${R}_bench_cfgA.json
cp_if $O/bench_cfgC.json $P/${R}_bench_cfgC.json
cp_if $O/bench_cam16.json $P/${R}_bench_cam16.json
cp_if $O/bench_vit.json $P/${R}_bench_vit.json
cp_if $O/bench_cfgB_train_bf16.json $P/${R}_bench_cfgB_train_bf16.json
cp_if $O/bench_cfgB_train_f32.json $P/${R}_bench_cfgB_train_f32.json
cp_if $O/f32_kernel_stats.csv $P/${R}_bench_cfgB_fp32_kernel_stats.csv
cp_if $O/bf16_kernel_stats.csv $P/${R}_bench_cfgB_bf16_kernel_stats.csv
cp_if $O/train_bf16_kernel_stats.csv $P/${R}_bench_cfgB_train_bf16_kernel_stats.csv
cp_if $O/train_f32_kernel_stats.csv $P/${R}_bench_cfgB_train_f32_kernel_stats.csv
cp_if $O/vit_bf16_kernel_stats.csv $P/${R}_vit_small16_adapter_b512_bf16_kernel_stats.csv
cp_if $O/vit_f32_kernel_stats.csv $P/${R}_vit_small16_adapter_b512_fp32_kernel_stats.csv
cp_if $O/measured_errors.txt $P/${R}_measured_errors.txt
cp_if $O/varlen_1k_bf16_kernel_stats.csv $P/${R}_varlen_64x1000_d384_bf16_kernel_stats.csv
cp_if $O/varlen_8k_f32_kernel_stats.csv $P/${R}_varlen_16x8192_d384_fp32_kernel_stats.csv
cp_if $O/varlen_bench.md $P/${R}_varlen_bench.md
cp_if $O/vit_mfma_pmc.txt $P/${R}_vit_mfma_pmc.txt
cp_if $O/traffic.txt $P/${R}_attn_traffic_cfgB.txt
cp_if $O/traffic/attn_traffic.json $P/${R}_attn_traffic_cfgB_bf16.json
cp_if $O/traffic_x3p_cfgB/attn_traffic_cfgB_fp32.json $P/${R}_attn_traffic_cfgB_fp32.json
cp_if $O/traffic_x3p_cfgC/attn_traffic_cfgC_fp32.json $P/${R}_attn_traffic_cfgC_fp32.json
cp_if $O/traffic_x3p_cfgA/attn_traffic_cfgA_fp32.json $P/${R}_attn_traffic_cfgA_fp32.json
cp_if $O/traffic_x3p_cfgB.txt $P/${R}_attn_traffic_cfgB_fp32.txt
cp_if $O/traffic_x3p_cfgC.txt $P/${R}_attn_traffic_cfgC_fp32.txt
cp_if $O/attn_x3p_pmc_sq.txt $P/${R}_attn_x3_pmc_sq.txt
cp_if $O/attn_x3p_timing.txt $P/${R}_attn_x3p_timing.txt
cp_if $O/gemm_hl_splitk.txt $P/${R}_gemm_hl_splitk.txt
cp_if $O/attn_mfma_pmc_sq.txt $P/${R}_attn_mfma_pmc_sq.txt
cp_if $O/attn_x3_timing.txt $P/${R}_attn_x3_timing.txt
cp_if $O/gemm_bench.txt $P/${R}_gemm_bench.txt
cp_if $O/gemm_x3_bench.txt $P/${R}_gemm_x3_bench.txt
cp_if $O/topk_bench.txt $P/${R}_topk_bench.txt
cp_if $O/sweep.md $P/${R}_sweep_N_D.md
for w in readme_dino_scratch readme_dino_adapter readme_mae_adapter; do cp_if $O/bench_$w.json $P/${R}_bench_$w.json; done
cp_if $O/cfgA_f32_kernel_stats.csv $P/${R}_bench_cfgA_fp32_kernel_stats.csv
cp_if $O/cfgC_f32_kernel_stats.csv $P/${R}_bench_cfgC_fp32_kernel_stats.csv
cp_if $O/readme_mae_f32_kernel_stats.csv $P/${R}_bench_readme_mae_adapter_fp32_kernel_stats.csv
cp_if $O/readme_scratch_f32_kernel_stats.csv $P/${R}_bench_readme_dino_scratch_fp32_kernel_stats.csv
cp_if $O/exact_attn_bench.txt $P/${R}_exact_attn_bench.txt
ls -la $P | grep ${R}_ | wc -l
