"""tools/bench_gemm.py (GPU box): the hand-written NT GEMM (csrc/gemm_nt_kernel.hpp) against the library GEMM it replaces.

  python tools/bench_gemm.py --check          correctness + race screen on small / ragged / full shapes
  python tools/bench_gemm.py --perf           interleaved A / B rounds on the DiM-L/2 shapes (random operands), TFLOP/s
Numbers quoted in DESIGN.md section 3.5 come from here."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dimsum_amd import native  # noqa: E402


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(dtype)


def check_plain(M, N, K, dtype, reps=3):
    a, b = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2)
    ref = (a.double() @ b.double().t())
    got = native.gemm_nt(a, b)
    err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    same = all(torch.equal(native.gemm_nt(a, b), got) for _ in range(reps))
    lib = torch.mm(a, b.t(), out_dtype=torch.float32)
    lib_err = (lib.double() - ref).abs().max().item() / ref.abs().max().item()
    ok = err < 2e-6 * max(1.0, (K / 1024) ** 0.5) + 1e-7 and same
    print(f"plain {str(dtype)[6:]:9s} M={M:6d} N={N:5d} K={K:5d}  rel err {err:.2e} (library {lib_err:.2e})  repeatable {same}  {'ok' if ok else 'FAIL'}", flush=True)
    return ok


def check_bias(M, N, K, dtype):
    a, b = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2)
    bias = rnd((N,), torch.float32, 3)
    ref = a.double() @ b.double().t() + bias.double()
    got = native.gemm_nt(a, b, bias=bias)
    err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    ok = err < 3e-6
    print(f"bias  {str(dtype)[6:]:9s} M={M:6d} N={N:5d} K={K:5d}  rel err {err:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
    return ok


def check_gated(M, F, H, with_bias=True):
    """the split3 epilogue against the unfused pair (library GEMM + gated GeLU pass) and against float64"""
    x = rnd((M, H), torch.float32, 4)
    w12 = rnd((2 * F, H), torch.float32, 5, scale=H ** -0.5)
    bias = rnd((2 * F,), torch.float32, 6, scale=0.1) if with_bias else None
    x3 = native.split3_rows(x, left=True)
    w3 = native.split3_rows(w12, left=False)
    got = native.gemm_nt(x3, w3, bias=bias, epilogue="gated_split3")                 # (M, 3F) bf16 [hi | hi | lo]
    x12 = x.double() @ w12.double().t()
    if bias is not None:
        x12 = x12 + bias.double()
    ref = torch.nn.functional.gelu(x12[:, :F], approximate="tanh") * x12[:, F:]
    hi, hi2, lo = got[:, :F].double(), got[:, F:2 * F].double(), got[:, 2 * F:].double()
    err = ((hi + lo) - ref).abs().max().item() / ref.abs().max().item()
    unf = native.gated_gelu_fwd(torch.mm(x3, w3.t(), out_dtype=torch.float32), bias, split3=True)
    uerr = ((unf[:, :F].double() + unf[:, 2 * F:].double()) - ref).abs().max().item() / ref.abs().max().item()
    ok = err < 2e-5 and torch.equal(hi, hi2)
    print(f"gated split3 M={M:6d} F={F:5d} H={H:5d} bias={with_bias}  rel err vs float64 {err:.2e} (unfused pair {uerr:.2e})  {'ok' if ok else 'FAIL'}", flush=True)
    # fp16 image
    s = 8.0
    x16, w16 = x.half(), w12.half()
    g16 = native.gemm_nt(x16, w16, bias=bias, epilogue="gated_f16", out_scale=s)
    x12h = x16.double() @ w16.double().t()
    if bias is not None:
        x12h = x12h + bias.double()
    refh = torch.nn.functional.gelu(x12h[:, :F], approximate="tanh") * x12h[:, F:]
    e16 = (g16.double() / s - refh).abs().max().item() / refh.abs().max().item()
    ok16 = e16 < 1e-3
    print(f"gated f16    M={M:6d} F={F:5d} H={H:5d}               rel err vs float64 (same operands) {e16:.2e}  {'ok' if ok16 else 'FAIL'}", flush=True)
    return ok and ok16


def timed(fn, n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


def perf(shapes, rounds, inner):
    out = []
    for name, M, N, K, dtype in shapes:
        a, b = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, scale=K ** -0.5)
        c = torch.empty((M, N), device="cuda", dtype=torch.float32)
        arms = {"library": lambda: torch.mm(a, b.t(), out=c) if False else torch.mm(a, b.t(), out_dtype=torch.float32),
                "gemm_nt": lambda: native.gemm_nt(a, b, out=c)}
        if N % 8 == 0:
            h3 = torch.empty((M, 3 * (N // 2)), device="cuda", dtype=torch.bfloat16)
            h16 = torch.empty((M, N // 2), device="cuda", dtype=torch.float16)
            if dtype == torch.bfloat16:
                arms["gemm_nt+gate->split3"] = lambda: native.gemm_nt(a, b, epilogue="gated_split3", out=h3)
                arms["library+gate pass"] = lambda: native.gated_gelu_fwd(torch.mm(a, b.t(), out_dtype=torch.float32), None, split3=True)
            else:
                arms["gemm_nt+gate->f16"] = lambda: native.gemm_nt(a, b, epilogue="gated_f16", out=h16)
        for f in arms.values():
            f()
        torch.cuda.synchronize()
        res = {k: [] for k in arms}
        for _ in range(rounds):
            for k, f in arms.items():
                res[k].append(timed(f, inner))
        fl = 2.0 * M * N * K
        row = {"shape": name, "M": M, "N": N, "K": K, "dtype": str(dtype)[6:]}
        for k, v in res.items():
            v.sort()
            med = v[len(v) // 2]
            row[k] = {"ms_median": round(med, 4), "ms_min": round(v[0], 4), "TF_median": round(fl / med / 1e9, 1)}
        print(json.dumps(row), flush=True)
        out.append(row)
    return out


def tune(M, N, K, dtype, rounds, inner, check=True):
    """interleaved rounds of the tuning variants of the plain fp32-output kernel (dimsum_gemm_params_t.tune_*)"""
    a, b = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, scale=K ** -0.5)
    c = torch.empty((M, N), device="cuda", dtype=torch.float32)
    arms = {"library": lambda: torch.mm(a, b.t(), out_dtype=torch.float32)}
    if N % 16 == 0 and dtype == torch.bfloat16:
        h3 = torch.empty((M, 3 * (N // 2)), device="cuda", dtype=torch.bfloat16)
        for gm in (2, 4, 8):
            arms[f"gate->split3 gm{gm}"] = (lambda gm=gm: native.gemm_nt(a, b, epilogue="gated_split3", out=h3, tune=(0, gm, 0)))
        h16 = torch.empty((M, N // 2), device="cuda", dtype=torch.float16)
        for gm in (2, 4, 8):
            arms[f"gate->f16 gm{gm}"] = (lambda gm=gm: native.gemm_nt(a, b, epilogue="gated_f16", out=h16, tune=(0, gm, 0)))
    variants = [("ship gm4", (0, 4, 0)), ("ship gm8", (0, 8, 0)), ("ship gm2", (0, 2, 0)), ("plain-stores gm8", (100, 8, 0)), ("no-epilogue", (2, 8, 0))]
    for name, t in variants:
        if dtype != torch.bfloat16 and t[0] != 0:
            continue
        arms[name] = (lambda t=t: native.gemm_nt(a, b, out=c, tune=t))
    if check:
        ref = torch.mm(a, b.t(), out_dtype=torch.float32)
        for name, f in arms.items():
            if "no-epi" in name or "gate" in name:
                continue
            c.zero_()
            r = f()
            if not torch.allclose(r, ref, rtol=1e-4, atol=1e-4 * ref.abs().max().item()):
                print("MISMATCH", name, (r - ref).abs().max().item(), flush=True)
    for f in arms.values():
        f()
    torch.cuda.synchronize()
    res = {k: [] for k in arms}
    for _ in range(rounds):
        for k, f in arms.items():
            res[k].append(timed(f, inner))
    fl = 2.0 * M * N * K
    print(f"tune M={M} N={N} K={K} {dtype}", flush=True)
    for k, v in res.items():
        v.sort()
        med = v[len(v) // 2]
        print(f"  {k:22s} median {med:7.4f} ms  min {v[0]:7.4f}  {fl / med / 1e9:7.1f} TF", flush=True)


def tiles_ab(rounds, inner, stagger=()):
    """the scaled-fp16 launches of one DiM-L/2 forward at batch 256 on 256-row tiles (tune 513) against 128-row tiles (tune 512: two
    4-wave workgroups per CU, csrc/gemm_nt_kernel.hpp kVarM128), interleaved rounds on one box; epilogues as the model runs them"""
    M = 65536
    def f16(shape, seed, scale=1.0):
        return native.rows_f16s(rnd(shape, torch.float32, seed, scale))
    shapes = [("in_proj (d-major out)", 2048, M, 512, "f32"), ("qkv + bias", M, 1536, 512, "bias"), ("proj + residual", M, 1024, 1024, "res"),
              ("w12 + gate -> f16", M, 8192, 1024, "gated"), ("w3 + gate residual", M, 1024, 4096, "gres"), ("xl512 w12 + gate", M, 9216, 1152, "gated"),
              ("xl512 in_proj", 2304, M, 576, "f32")]
    for name, m, n, k, kind in shapes:
        a, b = f16((m, k), 1), f16((n, k), 2, k ** -0.5)
        kw = dict(scales=(a.inv, b.inv))
        if kind == "bias":
            kw["bias"] = rnd((n,), torch.float32, 3)
        if kind in ("res", "gres"):
            kw.update(bias=rnd((n,), torch.float32, 3), residual=rnd((m, n), torch.float32, 4))
        if kind == "gres":
            kw.update(gate=rnd((m // 256, n), torch.float32, 5), rows_per_batch=256)
        if kind == "gated":
            w, l1 = native.rows_f16s(rnd((n, k), torch.float32, 2, k ** -0.5), want_l1=True)
            b12 = rnd((n,), torch.float32, 6, 0.1)
            kw.update(bias=b12, epilogue="gated_f16", gate_bound=torch.cat([l1 * (1 + 2.0 ** -10), b12.abs().max().reshape(1)]).contiguous())
        if kind == "gated":
            out = None
        else:
            out = torch.empty((m, n), device="cuda", dtype=torch.float32)
        arms = {"256-row tiles": lambda: native.gemm_nt(a.data, b.data, out=out, tune=(513, 0, 0), **kw),
                "128-row tiles": lambda: native.gemm_nt(a.data, b.data, out=out, tune=(512, 0, 0), **kw)}
        arms["persistent"] = lambda: native.gemm_nt(a.data, b.data, out=out, tune=(514, 0, 0), **kw)       # (where built: else the launch heuristics' choice)
        for st in stagger:          # first round of the odd CUs `st` x 1024 cycles late (gemm_nt_kernel.hpp cu_stagger)
            arms[f"256 stagger {st}"] = lambda st=st: native.gemm_nt(a.data, b.data, out=out, tune=(513, 0, st), **kw)
            arms[f"128 stagger {st}"] = lambda st=st: native.gemm_nt(a.data, b.data, out=out, tune=(512, 0, st), **kw)
        r0, r1, r2 = arms["256-row tiles"](), arms["128-row tiles"](), arms["persistent"]()
        same = (torch.equal(r0.data, r1.data) and torch.equal(r0.data, r2.data)) if kind == "gated" else (torch.equal(r0, r1) and torch.equal(r0, r2))
        torch.cuda.synchronize()
        res = {k_: [] for k_ in arms}
        for _ in range(rounds):
            for k_, f in arms.items():
                res[k_].append(timed(f, inner))
        row = {"shape": name, "M": m, "N": n, "K": k, "bit_identical": same}
        for k_, v in res.items():
            v.sort()
            row[k_] = {"ms_median": round(v[len(v) // 2], 4), "ms_min": round(v[0], 4), "TF_median": round(2.0 * m * n * k / v[len(v) // 2] / 1e9, 1)}
        print(json.dumps(row), flush=True)
        del a, b, out, kw, r0, r1
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--perf", action="store_true")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--inner", type=int, default=10)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--tune", action="store_true")
    ap.add_argument("--stagger", default="", help="--tiles: also time these start delays of the odd CUs (x 1024 cycles), e.g. 4,8,16 (a library built with -DDIMSUM_GEMM_TUNE: tools/scratch/build_variant.sh)")
    ap.add_argument("--tiles", action="store_true", help="A / B of the 256-row against the 128-row tile variant on the scaled-fp16 launch shapes")
    ap.add_argument("--pmc-run", action="store_true", help="a few launches of the w12-shape kernels (under rocprofv3 --pmc)")
    ap.add_argument("--pmc-run-f16s", action="store_true", help="a few launches of every scaled-fp16 GEMM launch class of the headline forward (w12 + gate persistent, w3 + gate residual, "
                                                                 "in_proj + conv, qkv -> fp16, out_proj TN over block-scaled out_z) and of the training TN with row factors (under rocprofv3 --pmc)")
    ap.add_argument("--pmc-run-tn", action="store_true", help="a few launches of the dW12-shape TN kernels (three-piece row stacks, pairs) and of the library's batched TN GEMM")
    args = ap.parse_args()
    ok = True
    if args.check:
        for dtype in (torch.bfloat16, torch.float16):
            for (M, N, K) in ((256, 256, 128), (256, 256, 192), (512, 512, 256), (512, 384, 320), (768, 1152, 1152), (256, 132, 128),
                              (1024, 8192, 3072)):
                ok &= check_plain(M, N, K, dtype)
        ok &= check_bias(512, 384, 256, torch.bfloat16)
        ok &= check_gated(512, 256, 128)
        ok &= check_gated(512, 1536, 384, with_bias=False)
        ok &= check_gated(1024, 4608, 1152)
        ok &= check_plain(65536, 8192, 3072, torch.bfloat16, reps=2) if not args.quick else True
        print("CHECK", "ok" if ok else "FAILED", flush=True)
    if args.pmc_run:
        a, b = rnd((65536, 3072), torch.bfloat16, 1), rnd((8192, 3072), torch.bfloat16, 2, scale=3072 ** -0.5)
        c = torch.empty((65536, 8192), device="cuda", dtype=torch.float32)
        h3 = torch.empty((65536, 3 * 4096), device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            native.gemm_nt(a, b, out=c)
            native.gemm_nt(a, b, epilogue="gated_split3", out=h3)
            native.gemm_nt(a, b, out=c, tune=(2, 8, 0))
            torch.mm(a, b.t(), out_dtype=torch.float32)
        torch.cuda.synchronize()
    if args.pmc_run_f16s:
        M = 65536
        f16 = lambda shape, seed, scale=1.0: native.rows_f16s(rnd(shape, torch.float32, seed, scale))
        x1024, x512, h4096 = f16((M, 1024), 1), f16((M, 512), 2), f16((M, 4096), 3)
        w12, l1 = native.rows_f16s(rnd((8192, 1024), torch.float32, 4, 1024 ** -0.5), want_l1=True)
        b12 = rnd((8192,), torch.float32, 5, 0.1)
        bound = torch.cat([l1 * (1 + 2.0 ** -10), b12.abs().max().reshape(1)]).contiguous()
        w3, win, wqkv = f16((1024, 4096), 6, 4096 ** -0.5), f16((2048, 512), 7, 512 ** -0.5), f16((1536, 512), 8, 512 ** -0.5)
        wq, lq = native.rows_f16s(rnd((1536, 512), torch.float32, 8, 512 ** -0.5), want_l1=True)
        bq = rnd((1536,), torch.float32, 9, 0.1)
        qb = torch.cat([lq * (1 + 2.0 ** -10), bq.abs().max().reshape(1)]).contiguous()
        res, gate, b3 = rnd((M, 1024), torch.float32, 10), rnd((256, 1024), torch.float32, 11), rnd((1024,), torch.float32, 12)
        cw, cb = rnd((1024, 4), torch.float32, 13), rnd((1024,), torch.float32, 14)
        oz = rnd((1024, M), torch.float16, 15, 1000.0)
        tab = torch.exp2(torch.randint(-20, -10, (M // 32, 16), device="cuda").float())
        wo_t = f16((512, 1024), 16, 1024 ** -0.5)
        wot, wo_inv = wo_t.data.t().contiguous(), wo_t.inv
        dy16, xx16 = f16((M, 8192), 17), x1024
        fac, cs = native.row_factors(dy16.inv, xx16.inv)
        dxz16 = native.rows_f16s(rnd((2048, M), torch.float32, 18))                                   # a d-major gradient: one scale per channel (long-row image kernel)
        for _ in range(3):
            native.gemm_nt(x1024.data, w12.data, bias=b12, epilogue="gated_f16", scales=(x1024.inv, w12.inv), gate_bound=bound)                                     # w12 + gate (persistent)
            native.gemm_nt(h4096.data, w3.data, bias=b3, scales=(h4096.inv, w3.inv), residual=res, gate=gate, rows_per_batch=256)                                   # w3 + gate residual
            native.gemm_nt(win.data, x512.data, scales=(win.inv, x512.inv), conv=(cw, cb, 256))                                                                      # in_proj + conv (m128)
            native.gemm_nt(x512.data, wq.data, bias=bq, epilogue="f16_qkv", scales=(x512.inv, wq.inv), gate_bound=qb, rows_per_batch=256, q_cols=512)                # qkv -> fp16
            native.gemm_tn(oz, wot, scales=(tab, wo_inv))                                                                                                            # out_proj TN rebase
            native.gemm_tn(dy16.data, xx16.data, row_scales=(fac, cs))                                                                                               # training dW12 (row factors)
            native.gemm_nn(dxz16.data, dxz16.inv, x512.data, x512.inv)                                                                                               # training d in_proj.weight (mixed layout)
        torch.cuda.synchronize()
    if args.pmc_run_tn:
        M = 65536
        dy3, x3 = rnd((3 * M, 8192), torch.bfloat16, 1), rnd((3 * M, 1024), torch.bfloat16, 2, scale=0.01)
        dyp, xp = native.PairImage(rnd((M, 2 * 8192), torch.bfloat16, 3)), native.PairImage(rnd((M, 2 * 1024), torch.bfloat16, 4, scale=0.01))
        for _ in range(3):
            native.gemm_tn(dy3, x3)
            native.gemm_tn_pairs(dyp, xp)
            torch.bmm(dy3.view(2, 3 * M // 2, 8192).transpose(1, 2), x3.view(2, 3 * M // 2, 1024), out_dtype=torch.float32).sum(0)
        torch.cuda.synchronize()
    if args.tune:
        tune(65536, 8192, 3072, torch.bfloat16, args.rounds, args.inner)
        tune(65536, 8192, 1024, torch.bfloat16, args.rounds, args.inner)
        tune(65536, 1024, 12288, torch.bfloat16, args.rounds, args.inner)
    if args.tiles:
        tiles_ab(args.rounds, args.inner, [int(v) for v in args.stagger.split(",") if v])
    if args.perf:
        shapes = [("w12 split3", 65536, 8192, 3072, torch.bfloat16),
                  ("w3 split3", 65536, 1024, 12288, torch.bfloat16),
                  ("in_proj split3", 65536, 2048, 3072, torch.bfloat16),
                  ("w12 fp16", 65536, 8192, 1024, torch.float16),
                  ("w3 fp16", 65536, 1024, 4096, torch.float16),
                  ("square 8192 bf16", 8192, 8192, 8192, torch.bfloat16)]
        if args.quick:
            shapes = shapes[:1]
        perf(shapes, args.rounds, args.inner)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
