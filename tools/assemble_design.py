"""Assemble DESIGN.md from docs_src/*.md + the still-valid sections of the round-3 text (profiles/history/DESIGN_rounds_1_3.md) and
fill the @@PLACEHOLDERS@@ from docs_src/numbers.json.   python tools/assemble_design.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
old = open(os.path.join(ROOT, "profiles/history/DESIGN_rounds_1_3.md")).read().split("\n")


def sect(first, last):        # 1-based inclusive line range of the round-3 file
    return "\n".join(old[first - 1:last])


def src(name):
    return open(os.path.join(ROOT, "docs_src", name)).read().rstrip("\n")


table = old[118:141]          # §4 intro + kernel table (lines 119-141)
rows = []
for line in table:
    if line.startswith("| **`sparse_attn_x3_kernel`**"):
        rows.append("| **`sparse_attn_x3p_kernel`** + `x3p_prep_kp` + `x3p_reduce` (sparse_attn_x3p.hip) | attention() in the fp32 path, dk = 128, 97 ≤ K ≤ 2048 "
                    "| MFMA (3 products per term) / issue-bound | @@TABLE_X3P@@ | round 4; see below |")
        rows.append("| `sparse_attn_x3_kernel` + `x3_reduce` (sparse_attn_x3.hip) | the same for dk = 64, K < 97, varlen batches, fp32-tensor inputs | MFMA / issue-bound "
                    "| 93–96 + 7 µs at config B (0.24) | round-3 kernel, operands split in registers |")
        rows.append("| **`gemm_hl_kernel`** (gemm.hip) | every fp32-class projection that fills the chip (Q\\|V with hl output, FFN-in, FFN-out + residual) | MFMA target "
                    "| Q\\|V 194–216 µs, FFN-in 397–433, FFN-out 386–423 → ≈ 1.1 PF/s issued; 79 % of the fp32 bag | see \"GEMM\" below |")
    elif line.startswith("| `critic_kernel<ln>`"):
        rows.append(line.replace("the histogram adds 0–1 µs |", "the histogram adds 0–1 µs; fp32 path (round 4): scores + the affine-free hl image `xhat` in one pass 41 µs (200 MB: 4.9 TB/s) |"))
    else:
        rows.append(line)
layout = sect(105, 116).replace(
    "| bf16 path | `xhat`",
    "| fp32 path, round 4 | `[Q \\| V]` [N, 4D] bf16 **hl image** (no fp32 Q / V tensor) | written by the Q\\|V projection's epilogue (`gemm_hl` `OUT = 3`), "
    "streamed by `sparse_attn_x3p_kernel` with LDS-DMA: head a, true column c of Q at byte `2·(2 a dk + 64 (c / 32) + c % 32)` (hi), `+ 64` (lo); V behind Q at column 2D |\n"
    "| | ONE normalised image `xhat` [N, 2D] bf16 hl (no affine) for both sublayers | as in the bf16 path: γ / β of LN0 / LN1 folded into Wq\\|Wv and W1 (in fp64, rounded once), written by the critic pass on its one read of the bag; after the attention the K patched rows are re-normalised into it (`FP32_SHARED_NORM`; needs equal eps) |\n"
    "| | Kp as the attention kernel's **fragment image** [h][⌈K/32⌉][dk/16][hi \\| lo][64 lanes] × 16 B | written by the key projection (`snf_linear_rows_x3_kpfrag_f32`), scaled by 1/√dk · log2 e |\n| bf16 path | `xhat`", 1)
parts = [src("00_head.md"), "", sect(52, 104), "", layout, "",
         "## 4. Kernels, rooflines, algorithmic bytes (§8d)", "", "\n".join(rows), "",
         src("41_x3p.md"), "", src("42_rest.md"), "", src("50_meas.md"), "", sect(699, 774), src("80_scope.md"), ""]
text = "\n".join(parts)
nums = json.load(open(os.path.join(ROOT, "docs_src", "numbers.json")))
for k, v in nums.items():
    text = text.replace("@@" + k + "@@", v)
left = sorted(set(re.findall(r"@@[A-Z0-9_]+@@", text)))
if left:
    print("unfilled:", left)
open(os.path.join(ROOT, "DESIGN.md"), "w").write(text)
print(len(text.split("\n")), "lines")
