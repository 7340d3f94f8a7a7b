#!/usr/bin/env python3
"""Headline benchmark: slides/sec of the Snuffy MIL aggregator + sparse-attention HBM GB/s on MI355X.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python bench.py --gpus N ...            (no launcher: re-executes itself as N ranks under torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

The headline leg is the reference's arithmetic (--precision fp32: fp32 tensors, products as split-bf16 x3 on the matrix cores);
the bf16 path is timed beside it (value_bf16 / roofline_bf16).  With N > 1 ranks an eval run also times training steps
(value_train): that is where the path's one collective, the flat-gradient RCCL all-reduce, lives.

A step = one bag (slide) through MILNet (critic -> top-Lambda -> sparse attention -> FFN -> head) per rank, bags already
resident in HBM.  Workload = BASELINE.json's metric shape (config B): N=32768 patches, D=768, h=6, Lambda=200.
Bag-parallel (weak scaling): every rank owns its own bags; --mode train adds backward + AdamW + ONE RCCL all-reduce of
the flat gradient per step.  Rank 0 prints one JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "cfgB": dict(N=32768, D=768, h=6, lam=200),    # BASELINE.json metric shape
    "cfgA": dict(N=8192, D=384, h=6, lam=200),
    "cfgC": dict(N=100000, D=768, h=6, lam=512),
    # BASELINE.json configs[4] / SURVEY 8(d) Cfg5: CAMELYON16-scale synthetic set, 400 slides, lengths
    # clip(round(lognormal(mu = ln 30000 - sigma^2/2, sigma = 0.5)), 1000, 100000), seed 0; sharded over the ranks (rank r takes
    # bags r::W), resident in HBM.  N below is only the nominal length (roofline micro-benchmark shape).
    "cam16": dict(N=30000, D=768, h=6, lam=200, bags=400),
    # the reference's own published CAMELYON16 recipes (reference README.md:609-669; SURVEY 8(d): "also report h=4, the README recipe"):
    # 4 heads (dk = 96 / 192), a random patch share, bags of the CAMELYON16 mean length.  The random share comes from the device
    # sampler (MILNet.configure(sampler="device"): the opt-in fast mode; value_reference_sampler = the same forward with the reference's
    # numpy draws on the host, the bit-exact parity mode)
    "readme_dino_scratch": dict(N=30000, D=384, h=4, lam=900, r=0.7777777777777778),
    "readme_dino_adapter": dict(N=30000, D=384, h=4, lam=500, r=0.5),
    "readme_mae_adapter": dict(N=30000, D=768, h=4, lam=500, r=0.5),
}
# BASELINE.json configs[3]: compute_feats.py's DINO ViT-S/16 + adapter (ffn_num 32, scalar 10) over 224 x 224 tiles, batch 512
VIT = dict(arch="vit_small", width=384, batch=512, img=224, patch=16, ffn_num=32, scalar="10")
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0


def build_net(D, h, lam, precision, device, r=0.0):
    from snuffy_amd.snuffy import build_milnet
    torch.manual_seed(0)
    net = build_milnet(D, h, "relu", lam, r, 1)
    for _, p in net.named_parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_normal_(p)
    for m in net.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.zeros_(m.bias)
    return net.to(device).configure(precision=precision, return_attention=False)


def small_bag_leg(device, precision, n=1000, d=384, nbags=64, steps=10):
    """Small bags are launch-latency bound one at a time: slides/s of `nbags` bags of n patches, one bag per
    forward (graph replay) against MILNet.forward_bags (all of them in one set of launches, graph replay)."""
    net = build_net(d, 6, 200, precision, device).eval()
    g = torch.Generator().manual_seed(77)
    bags = [torch.randn(1, n, d, generator=g).to(device) for _ in range(nbags)]
    net.configure(graph_max_patches=1 << 20)
    with torch.no_grad():
        per_bag = nbags / (timed(lambda: [net(x) for x in bags], steps) * 1e-3)
        packed = nbags / (timed(lambda: net.forward_bags(bags), steps) * 1e-3)
    return {"workload": "%d bags x %d patches, D=%d, h=6, Lambda=200, eval forward, %s" % (nbags, n, d, precision),
            "slides_per_s_one_bag_per_forward": round(per_bag, 1), "slides_per_s_packed": round(packed, 1),
            "what": "MILNet.forward_bags: segmented top-k / attention / head kernels over the packed rows, same selections"}


def timed(fn, iters, warmup=2):
    """Average ms per call measured with HIP events on torch's current stream (the stream our kernels launch on)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_rooflines(wl, precision, device, wl_name="cfgB"):
    """Live timing of the north-star kernels on synthetic operands of the workload's shape."""
    from snuffy_amd import functional as SF
    from snuffy_amd import ops
    N, D, h, lam = wl["N"], wl["D"], wl["h"], wl["lam"]
    K, dk = min(lam, N), D // h
    D_true = D
    if SF.head_pad(dk) not in (None, dk) and (precision == "fp32" or ops.mfma_attn_supported(K, SF.head_pad(dk))):
        # head widths between the kernel's ride zero-padded (functional.head_pad: the README recipes' dk = 96 -> 128): the launch
        # streams the padded Q | V image; algorithmic bytes below stay those of the TRUE width
        dk = SF.head_pad(dk)
        D = h * dk
    g = torch.Generator(device="cpu").manual_seed(7)
    dt = torch.bfloat16 if precision == "bf16" else torch.float32
    elt = 2 if precision == "bf16" else 4
    out = {}
    scores = torch.randn(N, generator=g).to(device)
    k_top = max(1, math.ceil(lam * (1.0 - wl.get("r", 0.0))))         # the critic's share of the K selected rows
    t_topk_alone = timed(lambda: ops.topk(scores, k_top), 20)
    # the selection as the model dispatches it: the first radix digit is counted inside the critic pass (fused selector), so
    # top-Lambda costs the select launch plus whatever the histogram adds to the critic -- timed as
    # (critic + histogram -> select) minus (critic alone) on a resident bag
    xb = torch.randn(N, D_true, generator=g).to(device)
    wc = (torch.randn(1, D_true, generator=g) / math.sqrt(D_true)).to(device)
    bc = torch.zeros(1, device=device)
    eps = 1e-5 if precision == "bf16" else None        # the bf16 model's critic pass also emits the normalised bf16 copy
    t_topk, fused = t_topk_alone, False
    if ops.critic_select(xb, wc, bc, eps) is not None:
        ops.topk(ops.critic_select(xb, wc, bc, eps)[0].view(-1), k_top)
        t_crit = timed((lambda: ops.critic_ln(xb, wc, bc, eps)) if eps is not None else (lambda: ops.critic(xb, wc, bc)), 20)
        t_both = timed(lambda: ops.topk(ops.critic_select(xb, wc, bc, eps)[0].view(-1), k_top), 20)
        t_topk, fused = max(t_both - t_crit, 0.0), True
    # gather of the selected rows + row -> slot map + key projection, as the model dispatches them (the unit's time has to
    # include them).  fp32-class with the pipelined attention: ONE launch (round 6); otherwise gather_slot_map + the K-row projection
    idx_sel = ops.topk(scores, K)
    wk_t = (torch.randn(D_true, D_true, generator=g) / math.sqrt(D_true)).to(device)
    gk_what = "gather_slot_map + key projection (two launches)"
    if (precision == "fp32" and SF.X3_HL_ATTENTION and SF.FP32_GEMM == "x3" and SF.X3_HL_KPFRAG and SF.X3_HL_KPFRAG_GATHER and dk == D_true // h
            and ops.x3_hl_attn_supported(K, dk) and ops.x3_hl_kpfrag_supported(K, h, dk) and D_true % 16 == 0):
        def gather_kproj():
            ops.gather_linear_rows_x3_kpfrag(xb, idx_sel, wk_t, None, h)
        gk_what = "gather + row -> slot map + key projection -> fragment image (one launch)"
    elif ops.linear_rows_x3_supported(K, D_true, D_true):
        def gather_kproj():
            xs_, _ = ops.gather_slot_map(xb, idx_sel)[:2]
            ops.linear_rows_x3(xs_, wk_t, None, out_dtype=dt)
    else:
        def gather_kproj():
            xs_, _ = ops.gather_slot_map(xb, idx_sel)[:2]
            torch.mm(xs_, wk_t.t())
    t_gk = timed(gather_kproj, 20, warmup=3)
    del xb, wk_t
    kp = torch.randn(K, D, generator=g).to(device)
    kp_in = kp.to(dt)   # the bf16 path hands the kernel a bf16 Kp (output of the bf16 key projection), as the model does
    # rotate over several operand sets so the 256 MiB Infinity Cache cannot serve the re-reads
    nset = max(2, int(math.ceil(600e6 / (2 * N * D * elt))))
    # Q and V as the model produces them: the two column halves of one fused projection output [N, 2D]
    qvs = [torch.randn(N, 2 * D, generator=g).to(device).to(dt) for _ in range(nset)]
    state = {"i": 0}
    if precision == "bf16" and ops.mfma_attn_supported(K, dk):   # what functional.encoder_layer dispatches for this precision
        def attn():
            i = state["i"] = (state["i"] + 1) % nset
            ops.sparse_attn_fwd_mfma(qvs[i][:, :D], qvs[i][:, D:], kp_in, N, h)
        kern = "sparse_attn_mfma_kernel+reduce_partials_kernel"
    elif precision == "fp32" and SF.X3_HL_ATTENTION and SF.FP32_GEMM == "x3" and ops.x3_hl_attn_supported(K, dk):
        # what functional.encoder_layer dispatches for a bag this size: the pipelined split-bf16 x3 kernel on the hl image the
        # Q | V projection writes (4 bytes per element, like the fp32 tensor); three launches: Kp split, main, reduction
        imgs = [ops.split_hl_rows(qv.float()) for qv in qvs]      # [N, 4 D] bf16 = image of Q | image of V
        kern = "sparse_attn_x3p_kernel+x3p_prep_kp_kernel+x3p_reduce_kernel"
        kp_x3p = kp
        if SF.X3_HL_KPFRAG and ops.x3_hl_kpfrag_supported(K, h, dk):
            # ... and the key projection in front writes Kp as the kernel's fragment image (no prep launch): two launches
            wk = (torch.randn(D, D, generator=g) / math.sqrt(D)).to(device)
            kp_x3p = ops.linear_rows_x3_kpfrag(torch.randn(K, D, generator=g).to(device), wk, None, h, scale=1.0 / math.sqrt(D_true // h))
            kern = "sparse_attn_x3p_kernel+x3p_reduce_kernel"
            del wk

        def attn():
            i = state["i"] = (state["i"] + 1) % nset
            ops.sparse_attn_fwd_x3_hl(imgs[i][:, :2 * D], imgs[i][:, 2 * D:], kp_x3p, h)
        elt = 4
    elif precision == "fp32" and ops.x3_attn_supported(K, dk):   # the fp32 path's kernel: split-bf16 x 3 on the matrix cores
        vs = [qv[:, D:].float().contiguous() for qv in qvs]
        qf = [qv[:, :D].float().contiguous() for qv in qvs]

        def attn():
            i = state["i"] = (state["i"] + 1) % nset
            ops.sparse_attn_fwd_x3(qf[i], vs[i], kp, h)
        kern = "sparse_attn_x3_kernel+x3_reduce_kernel"
        elt = 4
    else:
        vs = [qv[:, D:].float().contiguous() for qv in qvs]
        qf = [qv[:, :D].float().contiguous() for qv in qvs]

        x3u = precision == "fp32" and ops.x3u_attn_supported(K, dk)   # functional.encoder_layer's dispatch for such head widths

        def attn():
            i = state["i"] = (state["i"] + 1) % nset
            (ops.sparse_attn_fwd_x3u(qf[i], kp, vs[i], h) if x3u else ops.sparse_attn_fwd(qf[i], kp, vs[i], h))
        # head widths outside the pipelined kernels: scores + softmax in split-bf16 x 3 (round 5) or exact on the f32 matrix cores,
        # P^T V exact on the f32 matrix cores; the vector-ALU scores kernel where neither applies
        ptv = "pt_v_lds_kernel" if K % 4 == 0 and dk % 4 == 0 else "pt_v_mfma_kernel"       # (operands staged through LDS where rows are 16-byte multiples)
        kern = ("scores_softmax_x3u_kernel+%s+reduce_slices_kernel" % ptv if x3u else
                "scores_softmax_mfma_kernel+%s+reduce_slices_kernel" % ptv if dk % 8 == 0 and K <= 1024 else
                "scores_softmax_kernel+%s+reduce_slices_kernel" % ptv)
        elt = 4
    # 20 launches over rotating operand sets: cold operands (every set is evicted from the 256 MiB Infinity Cache before it comes
    # back).  The bf16 kernel takes the same time inside the bag pipeline (rocprofv3: 37 + 7 us); the fp32-class kernel is faster
    # there (89 + 7 us against 105-114 us here), where Q | V were written by the projection just before -- see after_producer.
    t_attn = timed(attn, 20, warmup=3)
    # the same launch right behind a producer of its operands (the role of the Q|V projection in the bag: Q and V were just
    # written and are largely still in the 256 MiB Infinity Cache) -- reported beside the cold figure, never instead of it
    t_warm = None
    if kern.startswith("sparse_attn_"):
        src = qvs[0] if precision == "bf16" else qvs[0].float()
        buf = torch.empty_like(src)
        if precision == "bf16":
            def run_warm():
                ops.sparse_attn_fwd_mfma(buf[:, :D], buf[:, D:], kp_in, N, h)
        elif kern.startswith("sparse_attn_x3p"):
            src = imgs[0]
            buf = torch.empty_like(src)

            def run_warm():
                ops.sparse_attn_fwd_x3_hl(buf[:, :2 * D], buf[:, 2 * D:], kp_x3p, h)
        else:
            def run_warm():
                ops.sparse_attn_fwd_x3(buf[:, :D], buf[:, D:], kp, h)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        acc = 0.0
        buf.copy_(src)
        for it in range(13):
            buf.mul_(1.0)          # producer: rewrites the operands in place (reads what it writes, like a projection's epilogue stream)
            e0.record()
            run_warm()
            e1.record()
            torch.cuda.synchronize()
            if it >= 3:
                acc += e0.elapsed_time(e1)
        t_warm = acc / 10
        del buf, src
    # algorithmic bytes: read Q and V once, read Kp, write O;  top-k: read N scores, write K indices -- at the TRUE width
    streamed_width = D
    D = D_true
    b_attn = 2 * N * D * elt + K * D * elt + K * D * 4
    b_topk = 4 * N + 8 * K
    b_gather = 2 * K * D * 4
    # HBM-side bytes per launch from the PMC passes (FETCH_SIZE x2 on gfx950, WRITE_SIZE; tools/pmc_traffic.sh), recorded
    # for exactly this workload under profiles/ -- counters cannot be collected inside a timed run
    traffic = None
    import glob
    tfiles = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                           "r*_attn_traffic_%s_%s.json" % (wl_name, precision))))
    if tfiles:   # newest round's PMC record for this workload / precision
        with open(tfiles[-1]) as f:
            rec = json.load(f)
        if rec.get("kernel", kern) == kern:
            traffic = int(rec["total_bytes"])
    # the same kernels INSIDE the bag pipeline, from the newest committed rocprofv3 kernel-stats summary of this workload / precision
    # (profiles/r*_bench_<workload>_<precision>_kernel_stats.csv: AverageNs of every kernel of `kern`) -- the live figure above is
    # the dispatched unit on cold operands, this one is what the unit costs where the model runs it
    in_bag = None
    import csv
    sfiles = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                           "r*_bench_%s_%s_kernel_stats.csv" % (wl_name, precision))))
    if sfiles:
        with open(sfiles[-1]) as f:
            rows = list(csv.DictReader(f))
        parts = {}
        for kname in kern.split("+"):
            hit = [r for r in rows if kname + "<" in r["Name"] or kname + "(" in r["Name"]]
            if hit:
                calls = max(int(r["Calls"]) for r in hit)        # several instantiations of one kernel: per-bag total
                parts[kname] = round(sum(float(r["TotalDurationNs"]) for r in hit) / calls / 1e3, 2)
        if parts and kern.split("+")[0] in parts:
            in_bag = dict(recorded=True, source="committed file profiles/" + os.path.basename(sfiles[-1]) + " (rocprofv3 of an earlier run; "
                          "not measured in this run)", us_per_launch=round(sum(parts.values()), 2), kernels_us=parts,
                          frac_recorded=round(b_attn / (sum(parts.values()) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4))
    # `achieved` / `frac` price the launch at the bytes it has to move at ITS operand width (bf16 Q, V, Kp here).  SURVEY
    # section 8(d) prices the same unit at the reference's fp32 tensors (8ND + 8KD; with the selector 8ND + 4N + 16KD + 8K):
    # that figure is reported next to it as survey_8d_* -- same time, twice the bytes on the bf16 path.
    b_attn_8d = 8 * N * D + 8 * K * D
    b_unit_8d = 8 * N * D + 4 * N + 16 * K * D + 8 * K
    out["roofline"] = dict(bound="hbm", kernel=kern, achieved=round(b_attn / (t_attn * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS,
                           unit="GB/s", frac=round(b_attn / (t_attn * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), traffic=traffic,
                           us_per_launch=round(t_attn * 1e3, 2), algorithmic_bytes=b_attn,
                           flops=4 * N * K * D * (3 if kern.startswith("sparse_attn_x3") else 1), operand_dtype=precision,
                           head_width=D // h, head_width_streamed=streamed_width // h, survey_8d_bytes=b_attn_8d,
                           survey_8d_frac=round(b_attn_8d / (t_attn * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           traffic_source=("committed file profiles/" + os.path.basename(tfiles[-1]) + " (separate --pmc passes; not measured "
                                           "in this run)") if traffic is not None else None)
    if in_bag is not None:
        out["roofline"]["in_bag_rocprof"] = in_bag
    if t_warm is not None:
        out["roofline"]["after_producer"] = dict(us_per_launch=round(t_warm * 1e3, 2),
                                                 frac=round(b_attn / (t_warm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                 what="one launch timed alone right behind an in-place rewrite of its Q | V operands (as in "
                                                      "the bag, behind the projection; includes that launch's dispatch latency); achieved / "
                                                      "frac above are on cold, rotating operand sets")
    # the WHOLE unit of SURVEY 8(d)'s B_sa = 8ND + 4N + 16KD + 8K: select + gather / key projection + attention (round 6: the gather and
    # the key projection are timed too -- rounds 1-5 counted the gather's bytes but not its launches)
    t_unit = t_attn + t_topk + t_gk
    b_unit = b_attn + b_topk + b_gather
    out["roofline_topk_attn"] = dict(bound="hbm", achieved=round(b_unit / (t_unit * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS,
                                     unit="GB/s", frac=round(b_unit / (t_unit * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     us_topk=round(t_topk * 1e3, 2), us_gather_kproj=round(t_gk * 1e3, 2), us_attn=round(t_attn * 1e3, 2),
                                     us_topk_standalone=round(t_topk_alone * 1e3, 2),
                                     topk="fused selector: (critic + histogram -> select) - critic" if fused else "one launch on the scores",
                                     gather_kproj=gk_what,
                                     frac_without_gather_kproj=round(b_unit / ((t_attn + t_topk) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     algorithmic_bytes=b_unit, survey_8d_bytes=b_unit_8d,
                                     survey_8d_frac=round(b_unit_8d / (t_unit * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    del qvs
    return out


def vit_leg(device, precision, steps, warmup, world=1, dist=None):
    """The extractor of BASELINE.json configs[3] on synthetic tiles already resident in HBM: images / s over all ranks (every rank
    runs its own batches: tile batches are independent, no collective), with the MFMA roofline of the block contractions."""
    from snuffy_amd import vit
    torch.manual_seed(0)
    w = VIT["width"]
    model = getattr(vit, VIT["arch"])(patch_size=VIT["patch"], adapter_ffn_scalar=VIT["scalar"], adapter_ffn_num=VIT["ffn_num"],
                                      adapter_d_model=w).to(device).eval().configure(precision)
    x = torch.rand(VIT["batch"], 3, VIT["img"], VIT["img"], device=device)
    for _ in range(warmup):
        model(x)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model(x)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    T = (VIT["img"] // VIT["patch"]) ** 2 + 1
    flops_img = 12 * (2 * T * w * 3 * w + 4 * T * T * w + 2 * T * w * w + 16 * T * w * w + 4 * T * w * VIT["ffn_num"]) \
        + 2 * (T - 1) * 3 * VIT["patch"] ** 2 * w
    rate = world * steps * VIT["batch"] / el
    issued = 3 if precision == "fp32" else 1      # split-bf16 x3: three bf16 MFMA products per fp32 product
    return dict(value=round(rate, 1), unit="img/s", ms_per_batch=round(el / steps * 1e3, 3), steps=steps, batch=VIT["batch"],
                workload="ViT-S/16 + adapter (ffn_num 32), 224 x 224 synthetic tiles, batch %d per rank" % VIT["batch"],
                roofline=dict(bound="mfma", achieved=round(flops_img * rate / 1e12, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                              frac=round(flops_img * rate / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), traffic=None,
                              issued_bf16_tflops=round(issued * flops_img * rate / 1e12, 1),
                              note="model FLOPs (9.2 GFLOP / image) over the dense bf16 MFMA peak; fp32 issues three bf16 "
                                   "products per product"))


def cpu_baseline(wl, budget_s=25.0):
    """The CPU oracle (a torch-CPU port of the reference's op sequence; it materialises A like the reference) timed on this
    host's cores.  More threads is not faster on a many-core host (round 1: 0.79 slides/s on 128 threads vs 1.7 on 8): the
    thread count is swept and the best is reported with its count."""
    from oracle import snuffy_oracle as orc          # cpu_baseline leg: the only place bench.py touches oracle/
    N, D, h, lam, r = wl["N"], wl["D"], wl["h"], wl["lam"], wl.get("r", 0.0)
    cores = os.cpu_count() or 1
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or cores
    except Exception:
        pass
    net = build_net(D, h, lam, "fp32", "cpu")
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    x = torch.randn(N, D, generator=torch.Generator().manual_seed(1234))
    sweep = sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores} or {cores})
    per = budget_s / len(sweep)
    rates, total_n, t_all = {}, 0, time.perf_counter()
    with torch.no_grad():
        for thr in sweep:
            torch.set_num_threads(thr)
            orc.milnet_forward(x, sd, h, "relu", lam, r, 1)             # warm-up at this thread count
            t0, n = time.perf_counter(), 0
            while True:
                orc.milnet_forward(x, sd, h, "relu", lam, r, 1)
                n += 1
                el = time.perf_counter() - t0
                if el > per * 0.7 or n >= 8:
                    break
            rates[thr] = n / el
            total_n += n + 1
    best = max(rates, key=rates.get)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(value=round(rates[best], 4), unit="slides/s", cores=best, kind="port", cpu_model=cpu_model,
                physical_cores=cores, thread_sweep={str(k): round(v, 4) for k, v in rates.items()},
                sample="%d eval forwards (A materialised, as the reference does) of one synthetic bag N=%d D=%d in %.0f s: "
                       "oracle/snuffy_oracle.py, torch-CPU fp32, threads swept over %s, best = %d threads"
                       % (total_n, N, D, time.perf_counter() - t_all, sweep, best))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a steady-state region of ~1 s (a 30-step region of 16 ms still sits on the clock ramp: 1.87 k slides/s, 400 steps
    # 1.97 k, 2000 steps 2.05 k at config B; shorter regions also scatter by +-10 % with the GPU's power state)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="cfgB", choices=sorted(WORKLOADS) + ["vit"],
                    help="cfgB (default, BASELINE.json's metric shape) / cfgA / cfgC / cam16: the MIL aggregator; vit: the patch-embedding "
                         "extractor of configs[3] as the headline (the default run reports it beside the aggregator as vit_*)")
    # the headline arithmetic is the reference's: fp32 tensors (products on the matrix cores as split-bf16 x3, fp32-class);
    # the bf16 path (north_star's 1e-2 class) rides beside it as value_bf16 / roofline_bf16
    ap.add_argument("--precision", default="fp32", choices=["bf16", "fp32"])
    ap.add_argument("--mode", default="eval", choices=["eval", "train"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--preroll-s", dest="preroll_s", type=float, default=0.6,
                    help="untimed pre-roll (seconds of the same steps) in front of every timed region, after the --warmup steps")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="issue every kernel of the eval forward from Python instead of replaying HIP graphs "
                         "(MILNet.configure(graph_max_patches=...)).  Same kernels and bit-identical results either way; the "
                         "replay removes ~0.24 ms of host-side issue per bag: +50 %% and more for bags of <= 8k patches, "
                         "within 1 %% for config B on a fast host, +15 %% on a slow one")
    ap.add_argument("--headline-only", action="store_true",
                    help="time only --precision (default: the other precision and the A-materialising forward are timed as "
                         "well and reported as value_f32 / value_bf16 / value_with_attention_output)")
    ap.add_argument("--gemm-table", action="store_true",
                    help="apply snuffy_amd/tuning/gemm_gfx950.csv (library-GEMM selections; helps the training shapes, "
                         "nothing measurable for the eval forward)")
    ap.add_argument("--train-steps", type=int, default=None,
                    help="timed steps of the training leg that an N > 1 eval run adds (value_train: fwd + bwd + AdamW + the "
                         "flat-gradient RCCL all-reduce); default min(100, max(5, steps // 20))")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: run the same command line as N ranks of ONE node (one process per
        # GPU, RCCL over xGMI) and hand back its exit code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU "
                         "(python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...)"
                         % (args.gpus, world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: snuffy_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("SNF_BENCH_FORCE_DIST"):     # the env switch exercises the RCCL code path on one rank (tests)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # "nccl" == RCCL on ROCm

    if args.gemm_table:
        from snuffy_amd.gemm_tuning import use_pretuned_gemms
        use_pretuned_gemms()

    if args.workload == "vit":
        steps = min(args.steps, 50)
        leg = vit_leg(device, args.precision, steps, min(args.warmup, 5), world, dist)
        if rank == 0:
            other = "fp32" if args.precision == "bf16" else "bf16"
            line = {"metric": "images/sec", "value": leg["value"], "unit": "img/s", "n_gpus": world, "steps": steps,
                    "warmup": min(args.warmup, 5), "ms_per_step": leg["ms_per_batch"], "higher_is_better": True, "scaling": "weak",
                    "vs_baseline": None, "dtype": {"bf16": "bf16", "fp32": "f32"}[args.precision], "data": "synthetic",
                    "config": {"workload": leg["workload"], "parallelism": "tile batches x%d (independent, no collective)" % world},
                    "roofline": leg["roofline"]}
            if not args.headline_only and world == 1:
                o = vit_leg(device, other, max(3, steps // 3), 2)
                line["value_" + {"bf16": "bf16", "fp32": "f32"}[other]] = o["value"]
                line["roofline_" + {"bf16": "bf16", "fp32": "f32"}[other]] = o["roofline"]
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    wl = WORKLOADS[args.workload]
    N, D, h, lam = wl["N"], wl["D"], wl["h"], wl["lam"]
    r_share = wl.get("r", 0.0)
    # bags resident in HBM before the timed region; several distinct bags per rank, cycled
    if "bags" in wl:
        import numpy as np
        rs = np.random.RandomState(0)
        sigma = 0.5
        lens = np.clip(np.round(rs.lognormal(np.log(N) - sigma * sigma / 2, sigma, wl["bags"])), 1000, 100000).astype(int)
        # this rank's shard of the slide list (each rank times `steps` of them).  Round 6: length-aware -- the bags are
        # ranked by patch count and dealt in groups of `world` neighbours (snuffy_amd.balance.step_groups, the trainer's own rule):
        # step i of every rank is a bag of (nearly) the same length, so neither the per-step all-reduce of a training step nor the
        # barrier at the end of the timed region waits for one rank's long tail (round-robin lens[rank::world]: a step costs
        # 1.93x the mean bag at 8 ranks on this list, grouped 1.02x)
        from snuffy_amd import balance
        visit = balance.step_groups(lens, world, np.random.RandomState(1).permutation(len(lens)))
        rr = np.concatenate([np.arange(len(lens)), np.arange((-len(lens)) % world)])
        wl = dict(wl, length_balance={"rule": "length-ranked groups of %d (snuffy_amd/balance.py)" % world,
                                      "step_cost_max_over_mean": round(balance.imbalance(lens, visit, world, True), 4),
                                      "round_robin_would_be": round(balance.imbalance(lens, rr, world, True), 4)})
        mine = lens[visit.reshape(-1, world)[:, rank]]
        g = torch.Generator().manual_seed(1234 + rank)
        bags = [torch.randn(1, int(n_i), D, generator=g).to(device) for n_i in mine]
        nbags = len(bags)
    else:
        nbags = max(2, min(8, int(2.0e9 // (N * D * 4))))
        g = torch.Generator().manual_seed(1234 + rank)
        bags = [torch.randn(1, N, D, generator=g).to(device) for _ in range(nbags)]
    labels = [torch.tensor([float(i % 2)], device=device) for i in range(nbags)]

    def measure(precision, return_attention=False, steps=None, warmup=None, use_graph=None, mode=None, sampler="device"):
        """Times `steps` steps of one configuration (barrier + synchronize on both sides, max over ranks).
        Returns (elapsed seconds, per-rank seconds, launch mode)."""
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        use_graph = args.graph if use_graph is None else use_graph
        mode = args.mode if mode is None else mode
        net = build_net(D, h, lam, precision, device, r_share)
        net.configure(return_attention=return_attention, sampler=sampler)
        if r_share > 0 and sampler != "device":
            use_graph = False                  # the reference's numpy draws synchronise with the host on every bag
        launch = "eager"
        if mode == "train":
            from snuffy_amd.train import BagParallelStepper
            stepper = BagParallelStepper(net, world_size=world, dist=dist, device=device, precision=precision)

            def step(i):
                stepper.step(bags[i % nbags], labels[i % nbags])
        else:
            net.eval()
            if use_graph:
                with torch.no_grad():   # untimed set-up: large bags bind a graph to their buffer on the second sighting
                    ref = net(bags[0])                       # eager reference of one bag
                    net.configure(graph_max_patches=1 << 20)
                    try:
                        for _ in range(2):
                            for b in bags:
                                net(b)
                        out = net(bags[0])
                        # (with a random share every forward draws other rows: only the critic scores repeat)
                        same = torch.equal(out[0], ref[0]) and (r_share > 0 or torch.equal(out[1], ref[1]))
                    except Exception:
                        same = False
                    if not same or not getattr(net, "_graphs", None):   # replay unavailable: kernels issued from Python
                        net.configure(graph_max_patches=0)
                    else:
                        launch = "hip graph replay"
                torch.cuda.synchronize()

            def step(i):
                with torch.no_grad():
                    net(bags[i % nbags])

        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        # untimed pre-roll by TIME on top of the W warm-up steps: a short timed region (the driver times 20 steps = 26 ms) would
        # otherwise sit on the clock ramp of a GPU that was idle a moment ago (round 3: 758 slides/s over 20 steps against
        # 777-818 over 2000).  Same steps, same bags, not timed; the timed region below is still EXACTLY `steps` steps.
        t_pre = time.perf_counter()
        j = 0
        while time.perf_counter() - t_pre < args.preroll_s:
            for _ in range(8):
                step(warmup + j)
                j += 1
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        torch.cuda.synchronize()
        mine_s = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        per_rank = [mine_s]
        if dist is not None:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            allr = [torch.zeros(1, device=device, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(allr, torch.tensor([mine_s], device=device, dtype=torch.float64))
            per_rank = [float(v.item()) for v in allr]
        del net
        return elapsed, per_rank, launch

    elapsed, per_rank, launch = measure(args.precision)            # the headline leg: EXACTLY args.steps timed steps
    extra = {}
    if not args.headline_only:
        # the reference's own arithmetic (fp32) next to the bf16 headline (or the other way round), same bags, same loop
        other = "fp32" if args.precision == "bf16" else "bf16"
        steps_o = args.steps if args.mode == "eval" else max(3, args.steps // 3)
        e2, _, l2 = measure(other, steps=steps_o, warmup=min(args.warmup, 10))
        extra[other] = dict(elapsed=e2, steps=steps_o, launch=l2)
        if args.mode == "eval":
            # the reference's MILNet.forward always materialises A [1, h, N, K] (157 MB at config B); the trainer discards it
            # (train.py:830) and so does the headline -- this is the same forward with A written out
            e3, _, l3 = measure(args.precision, return_attention=True, warmup=min(args.warmup, 10))
            extra["with_A"] = dict(elapsed=e3, steps=args.steps, launch=l3)
            # the fp32 leg runs its projections as split-bf16 x3 products on the matrix cores (fp32-class: ~2^-17 per product,
            # fp32 accumulate); this is the same forward with plain fp32 library GEMMs instead
            from snuffy_amd import functional as SF
            keep = SF.FP32_GEMM
            SF.FP32_GEMM = "library"
            try:
                steps_l = max(5, args.steps // 2)
                e4, _, l4 = measure("fp32", steps=steps_l, warmup=min(args.warmup, 10))
            finally:
                SF.FP32_GEMM = keep
            extra["f32_library_gemm"] = dict(elapsed=e4, steps=steps_l, launch=l4)
    if r_share > 0 and args.mode == "eval" and not args.headline_only:
        # the same forward with the random share drawn by the reference's np.random.choice on the host (bit-exact parity mode)
        steps_r = max(5, args.steps // 4)
        e6, _, l6 = measure(args.precision, steps=steps_r, warmup=min(args.warmup, 10), sampler="reference")
        extra["reference_sampler"] = dict(elapsed=e6, steps=steps_r, launch=l6)
    train_leg = None
    if args.mode == "eval" and (dist is not None or not args.headline_only):   # at N = 1 too: one short leg, so that the
        # driver's single-GPU record also carries a training number (no collective at world 1: the stepper is the trainer)
        # the eval forward has no collective in its data path (bags are independent): an N-rank eval line says nothing about
        # the one exchange the north star names, the flat-gradient all-reduce.  The same ranks therefore also time training
        # steps (fwd + bwd + AdamW + ONE RCCL all-reduce of the flat fp32 gradient per step), reported as value_train.
        steps_t = args.train_steps or min(100, max(5, args.steps // 20))
        train_leg = {}
        for prec in ((args.precision,) if args.headline_only else (args.precision, "bf16" if args.precision == "fp32" else "fp32")):
            e5, pr5, _ = measure(prec, steps=steps_t, warmup=min(args.warmup, 5), mode="train")
            train_leg[prec] = dict(elapsed=e5, steps=steps_t, per_rank=pr5)

    if rank == 0:
        K = min(lam, N)
        flops_fwd = 20 * N * D * D + 4 * N * K * D + 4 * K * D * D
        dt_name = {"bf16": "bf16", "fp32": "f32"}
        line = {
            "metric": "slides/sec", "value": round(world * args.steps / elapsed, 3), "unit": "slides/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preroll_s": args.preroll_s,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dt_name[args.precision], "data": "synthetic",
            "config": {"workload": "%s: MILNet %s, 1 bag/step/rank, N=%d patches D=%d h=%d Lambda=%d (K=%d%s) depth=1 relu mlp x4"
                                   % (args.workload, "train step (fwd+bwd+AdamW+all-reduce)" if args.mode == "train" else
                                      "eval forward", N, D, h, lam, K,
                                      "" if r_share == 0 else ", random_patch_share %.4g: %d top + %d random rows, device sampler"
                                      % (r_share, math.ceil(lam * (1 - r_share)), int(lam * r_share))),
                       "parallelism": "bag-parallel x%d" % world, "bags_resident_per_rank": nbags,
                       "launch": launch, "rccl_ranks": world if dist is not None else 0,
                       "per_rank_slides_per_s": [round(args.steps / t, 2) for t in per_rank],
                       "rank_time_max_over_mean": round(max(per_rank) / (sum(per_rank) / len(per_rank)), 4),
                       "model_tflops_per_s": round(flops_fwd * (3 if args.mode == "train" else 1) * world * args.steps
                                                   / elapsed / 1e12, 2)},
        }
        if "length_balance" in wl:
            line["config"]["length_balance"] = wl["length_balance"]
        for key, rec in extra.items():
            if key == "with_A":
                line["value_with_attention_output"] = round(world * rec["steps"] / rec["elapsed"], 3)
            elif key == "reference_sampler":
                line["value_reference_sampler"] = round(world * rec["steps"] / rec["elapsed"], 3)
            elif key == "f32_library_gemm":
                line["value_f32_library_gemm"] = round(world * rec["steps"] / rec["elapsed"], 3)
            else:
                line["value_" + dt_name[key]] = round(world * rec["steps"] / rec["elapsed"], 3)
                line["ms_per_step_" + dt_name[key]] = round(rec["elapsed"] / rec["steps"] * 1e3, 4)
        if train_leg:
            for prec, rec in train_leg.items():
                sfx = "" if prec == args.precision else "_" + dt_name[prec]
                line["value_train" + sfx] = round(world * rec["steps"] / rec["elapsed"], 3)
                line["ms_per_step_train" + sfx] = round(rec["elapsed"] / rec["steps"] * 1e3, 4)
            line["train_leg"] = {"steps": train_leg[args.precision]["steps"], "rccl_ranks": world,
                                 "all_reduce_bytes_per_step": 4 * sum(p.numel() for p in build_net(D, h, lam, "fp32", "cpu", r_share).parameters()),
                                 "what": "fwd + bwd + fused AdamW + ONE flat-gradient all-reduce (sum, /W) per step, 1 bag per rank"}
        line["arithmetic"] = {
            "bf16": "bf16 MFMA operands, fp32 accumulate / softmax / LayerNorm / residual (north_star's 1e-2 class)",
            "f32": "fp32 tensors; products as split-bf16 x3 on the MFMA units (2^-17 per product, fp32 accumulate), "
                   "logits within 1e-5 of plain fp32 (north_star's 1e-3 class); value_f32_library_gemm = the same forward "
                   "with fp32 library GEMMs"}
        if not args.no_roofline:   # rank 0 only, after the timed region (the other ranks wait at the closing barrier)
            line.update(kernel_rooflines(wl, args.precision, device, args.workload))
            if not args.headline_only:
                other = "fp32" if args.precision == "bf16" else "bf16"
                for k, v in kernel_rooflines(wl, other, device, args.workload).items():
                    line[k + "_" + dt_name[other]] = v
        if world == 1 and not args.headline_only and args.workload == "cfgB":
            # BASELINE.json configs[3] beside the aggregator: the ViT-S/16 + adapter extractor, both arithmetics
            for prec, st in (("bf16", 8), ("fp32", 3)):
                leg = vit_leg(device, prec, st, 2)
                line["vit_" + dt_name[prec]] = {k: leg[k] for k in ("value", "unit", "ms_per_batch", "steps", "batch", "workload", "roofline")}
        if world == 1 and not args.headline_only and args.workload == "cfgB":
            line["small_bags"] = small_bag_leg(device, args.precision)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
