#!/usr/bin/env python3
"""Roofline of the hand-written kernels at SCALED shapes (at the BASELINE shapes every
non-GEMM kernel is launch-latency bound, so the HBM fraction is only meaningful when the problem is
scaled until the kernel runs for tens of microseconds).

Each case launches the kernel `reps` times back to back; durations come from HIP events recorded
inside libjorldy_hip around every launch (jh_prof_*), algorithmic bytes/flops from the per-unit
figures of SURVEY.md §8(d) / DESIGN.md.  Run it under `rocprofv3 --kernel-trace --stats` (and with
`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` in separate passes) to get the agreeing trace and the HBM
traffic; `--only NAME` restricts to one case so PMC passes stay short.

    python tools/roofline_scaled.py [--only gae] [--reps 20] [--out gpurun_out/roofline_scaled.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np
import torch

HBM_PEAK, MFMA_PEAK = 8000.0, 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from jorldy_amd import _lib as L
    from jorldy_amd import ops

    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    cases = []

    def case(name, kernel, bound, work_per_launch, fn, note=""):
        if args.only and not name.startswith(args.only):
            return
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ops.lib_profile(True)
        for _ in range(args.reps):
            fn()
        prof = ops.lib_profile_report()
        ops.lib_profile(False)
        kernels = kernel if isinstance(kernel, (list, tuple)) else [kernel]
        n, ms = 0, 0.0
        for k in kernels:
            if k not in prof:
                continue  # e.g. the totals kernel of the PPO loss only exists beyond 64 blocks
            n = max(n, prof[k][0])
            ms += prof[k][1]
        avg_s = ms / n * 1e-3
        if bound == "mfma":
            ach, peak, unit = work_per_launch / avg_s / 1e12, MFMA_PEAK, "TFLOP/s"
        else:
            ach, peak, unit = work_per_launch / avg_s / 1e9, HBM_PEAK, "GB/s"
        row = dict(case=name, kernel="+".join(kernels), bound=bound, avg_us=round(avg_s * 1e6, 2), work_per_launch=work_per_launch,
                   achieved=round(ach, 2), unit=unit, peak=peak, frac=round(ach / peak, 4), launches=n, note=note)
        cases.append(row)
        print(json.dumps(row), flush=True)

    # ---- GAE: 24 B / transition -------------------------------------------------------------------
    for W, T in ((8192, 128), (32, 2048), (64, 2048), (65536, 128)):  # (32, 2048): config.ppo.mujoco's own shape (rows > 256 steps: a workgroup per row)
        M = W * T
        r, v, vn = rnd(M, 1), rnd(M, 1), rnd(M, 1)
        d = (torch.rand(M, 1, device=dev, generator=g) < 0.02).float()
        case(f"gae_W{W}_T{T}", ["jh_gae_kernel", "jh_gae_long_kernel"], "hbm", 24.0 * M, lambda: ops.gae(r, d, v, vn, T, 0.99, 0.95, False), "24 B/transition, no standardise")
        case(f"gae_std_W{W}_T{T}", ["jh_gae_kernel", "jh_gae_long_kernel"], "hbm", 36.0 * M, lambda: ops.gae(r, d, v, vn, T, 0.99, 0.95, True), "24 + 12 B/transition (standardise re-reads hit L2)")
    # ---- PPO loss: discrete A=2 44 B/sample (+8 B idx); continuous A=3 88 B ---------------------------
    for B in (1 << 20,):
        A = 2
        z, vp = rnd(B, A), rnd(B, 1)
        act = torch.randint(0, A, (B, 1), device=dev, generator=g).float()
        adv, ret, vold, lpo = rnd(B, 1), rnd(B, 1), rnd(B, 1), -torch.rand(B, 1, device=dev, generator=g)
        case(f"ppo_loss_disc_B{B}", ["jh_ppo_fwd_kernel<CONT>", "jh_ppo_totals_kernel", "jh_ppo_bwd_kernel<CONT>"], "hbm", 44.0 * B + 32.0 * B,
             lambda: ops.ppo_loss_discrete(z, vp, None, act, adv, ret, vold, lpo, 0.1, 1.0, 0.01), "two-pass path: 44 B + the 8 reads of the recompute pass")
    # ---- gather: Atari-shaped uint8 rows and CartPole rows ------------------------------------------
    N = 50000
    st = ops.DeviceStore(N, [("state", L.JH_U8, 28224, (4, 84, 84)), ("next_state", L.JH_U8, 28224, (4, 84, 84)), ("reward", L.JH_F32, 3, (3, 1)), ("done", L.JH_U8, 3, (3, 1)), ("action", L.JH_I64, 1, (1,))])
    st.column("state").random_(0, 256, generator=g)
    st.column("next_state").random_(0, 256, generator=g)
    st.lib.jh_store_clear(st.h)
    # mark the ring full without copying 2.8 GB through the host
    import ctypes as C
    dummy = {k: st.column(k) for k in st.names}
    st.push_device(dummy, N)
    for B in (32, 4096):
        idx = torch.randint(0, N, (B,), device=dev, generator=g)
        rowb = 2 * 28224 + 12 + 3 + 8
        case(f"gather_atari_u8_B{B}", "jh_gather_kernel", "hbm", B * (rowb + rowb) + 8 * B, lambda: st.gather(idx, as_float=False), "uint8 kept until the conv: read + write 56.5 KB/sample")
        case(f"gather_atari_f32_B{B}", "jh_gather_kernel", "hbm", B * (rowb + 4 * 2 * 28224 + 4 * 7) + 8 * B, lambda: st.gather(idx), "as_tensor semantics: fp32 out (4x write)")
    del st
    N2 = 4 << 20
    st2 = ops.DeviceStore(N2, [("state", L.JH_F32, 4, (4,)), ("action", L.JH_I64, 1, (1,)), ("reward", L.JH_F32, 1, (1,)), ("next_state", L.JH_F32, 4, (4,)), ("done", L.JH_U8, 1, (1,))])
    st2.push_device({k: st2.column(k) for k in st2.names}, N2)
    B = 1 << 20
    idx = torch.randint(0, N2, (B,), device=dev, generator=g)
    case(f"gather_cartpole_B{B}", "jh_gather_kernel", "hbm", B * (45.0 + 44.0 + 8.0), lambda: st2.gather(idx), "random rows: 45 B read + 44 B fp32 written + idx")
    del st2
    # ---- PER: descent = ~20 dependent 8 B loads / sample ---------------------------------------------
    Np = 1 << 20
    tree = ops.SumTree(Np, 1e-3)
    pr = np.random.RandomState(0).rand(Np) ** 0.5
    for o in range(0, Np, 1 << 18):
        tree.push(1 << 18, pr[o : o + (1 << 18)])
    for B in (32, 65536):
        u = np.random.RandomState(1).rand(B)
        case(f"per_sample_B{B}", ["jh_per_sample_kernel", "jh_per_norm_kernel"], "hbm", B * (20 * 8 + 8 + 8 + 8 + 8 + 4.0), lambda: tree.sample(0.4, np.zeros(0, np.int64), u), "latency-bound dependent chain: 160 B/sample of tree + outputs")
    B = 2048
    ii = torch.randint(Np - 1, 2 * Np - 1, (B,), device=dev, generator=g)
    pp = torch.rand(B, device=dev, generator=g)
    case(f"per_update_B{B}", ["jh_per_delta_kernel", "jh_per_climb_kernel"], "hbm", B * 320.0, lambda: tree.update(ii, pp), "serial per-node fp64 chains (bit-exact order): not a bandwidth kernel")
    del tree
    # ---- TD / C51 ----------------------------------------------------------------------------------
    B, A = 1 << 20, 2
    q, qn, qt = rnd(B, A), rnd(B, A), rnd(B, A)
    a = torch.randint(0, A, (B,), device=dev, generator=g).float()
    r3, d3, w = rnd(B, 3), (torch.rand(B, 3, device=dev, generator=g) < 0.1).float(), torch.rand(B, device=dev, generator=g)
    case(f"td_per_nstep_B{B}", "jh_td_loss_kernel", "hbm", B * 4.0 * (3 * A + 1 + 6 + 1 + A + 1), lambda: ops.td_loss(q, qt, a, r3, d3, 0.99, q_next_online=qn, weights=w, alpha=0.6, n_step=3), "3A+8 reads, A+1 writes per sample")
    B, A, K = 65536, 4, 51
    lg, nl, tl = rnd(B, A, K), rnd(B, A, K), rnd(B, A, K)
    a = torch.randint(0, A, (B,), device=dev, generator=g).float()
    r3, d3, w = rnd(B, 3), (torch.rand(B, 3, device=dev, generator=g) < 0.1).float(), torch.rand(B, device=dev, generator=g)
    case(f"c51_rainbow_B{B}", "jh_c51_kernel", "hbm", B * 4.0 * (3 * A * K + A * K + 6 + 2 + 2), lambda: ops.c51_loss(lg, tl, a, r3, d3, -1, 10, 0.99, next_logit_online=nl, weights=w, alpha=0.5, n_step=3), "3 logit tensors read + grad tensor written")
    # ---- encoder GEMMs at scaled minibatches: whatever MFMA kernels the library runs for these calls, with the flops the
    # library itself declares per launch (jh_prof_*: the kernels are idempotent, repeated inside one event pair)
    def mfma_cases(tag, fn):
        if args.only and not tag.startswith(args.only):
            return
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ops.lib_profile(True, 20)
        for _ in range(args.reps):
            fn()
        prof = ops.lib_profile_report()
        ops.lib_profile(False)
        for k, (n, ms, work) in sorted(prof.items()):
            if work <= 0:
                continue
            avg_s = ms / n * 1e-3
            ach = work / n / avg_s / 1e12
            row = dict(case=f"{tag}:{k}", kernel=k, bound="mfma", avg_us=round(avg_s * 1e6, 2), work_per_launch=work / n, achieved=round(ach, 2),
                       unit="TFLOP/s", peak=MFMA_PEAK, frac=round(ach / MFMA_PEAK, 4), launches=n, note="fp32 MFMA 16x16x4; flops declared by the library")
            cases.append(row)
            print(json.dumps(row), flush=True)

    for Bm in (256, 1024, 2048, 8192):
        net = ops.PPONet(4, 512, 2, False, 16384, dev)
        net.params.normal_(0, 0.05, generator=g)
        x = rnd(Bm, 4)
        gz, gv = rnd(Bm, 2) / Bm, rnd(Bm, 1) / Bm
        mfma_cases(f"mlp_fwd_B{Bm}", lambda: net.forward(x))
        net.forward(x)
        mfma_cases(f"mlp_bwd_B{Bm}", lambda: net.backward(x, None, gz, None, gv))
        if Bm == 256:
            case(f"adam_B{Bm}", "jh_adam_kernel<false>", "hbm", 4.0 * net.n_params * 7, lambda: net.adam_step(1.0), "g r/w, p/m/v r+w: 28 B/param")
        del net
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(cases, f, indent=1)


if __name__ == "__main__":
    main()
