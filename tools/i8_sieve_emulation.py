#!/usr/bin/env python3
"""VERDICT r3 "next" 5, the gated experiment: would an 8-bit first stage of the sieve (v_mfma_i32_16x16x64_i8: half the bytes
and half the matrix cycles of the bf16 filter) keep its candidate lists small enough to pay?  Emulated on the CPU with the
RIGOROUS bound the shipped filter uses (Cauchy-Schwarz on what the index and the query actually lose to the rounding:
|x.q - x^.q^| <= |x - x^| |q| + |x^| |q - q^|, integer accumulation exact), on bench.py's three corpora, with the sieve's own
protocol: T0 from a 32K-row sample, launch 1 over the first 1/16 of the rows, T1 = the k-th largest lower bound of its
candidates, launch 2 over the rest, then the rows whose upper bound reaches the k-th largest lower bound are "evaluated"
(the reference formula in float64).

    python3 tools/i8_sieve_emulation.py [rows=10000000] [queries=16] [k=10]

Variants: bf16 (the shipped filter: control - compare with bench.py's `sieve` counters), i8 with one scale per index,
per 32-row tile, per row (a row / tile scale costs the filter a multiply per value or a bound per tile); `_rowm` = the same
with every row's OWN residual in the margin instead of the index's largest (what a per-row margin could reach at best).
Inner product on unit rows (squared L2 doubles values and margins alike: same lists)."""
import sys
import time

import numpy as np
import torch

CH = 1_000_000


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def quant_i8(x, scale):
    """x ~ scale * round(x / scale), clipped to +-127; returns the dequantised float32 image"""
    return torch.clamp(torch.round(x / scale), -127, 127) * scale


def corpus_chunks(kind, n, d):
    g = torch.Generator()
    if kind == "isotropic":
        for c0 in range(0, n, CH):
            g.manual_seed(1234 + c0 // CH)
            m = min(CH, n - c0)
            x = torch.randn((m, d), generator=g)
            yield c0, x / x.norm(dim=1, keepdim=True)
    elif kind == "clustered":
        g.manual_seed(31337)
        centres = torch.randn((4096, d), generator=g)
        centres /= centres.norm(dim=1, keepdim=True)
        corpus_chunks.centres = centres
        for c0 in range(0, n, CH):
            g.manual_seed(777 + c0 // CH)
            m = min(CH, n - c0)
            cid = torch.randint(0, 4096, (m,), generator=g)
            x = torch.randn((m, d), generator=g)
            x /= x.norm(dim=1, keepdim=True)
            x = 0.9486833 * centres[cid] + 0.3162278 * x
            yield c0, x / x.norm(dim=1, keepdim=True)
    else:  # near_duplicate: 10 % of the rows in groups of 32 within 1e-7 of the group's first
        for c0 in range(0, n, CH):
            g.manual_seed(1234 + c0 // CH)
            m = min(CH, n - c0)
            x = torch.randn((m, d), generator=g)
            x /= x.norm(dim=1, keepdim=True)
            g.manual_seed(4141 + c0 // CH)
            ng = int(m * 0.10) // 32
            starts = torch.randperm(m // 32, generator=g)[:ng] * 32
            idx = (starts[:, None] + torch.arange(32)[None, :]).reshape(-1)
            base = x[starts].repeat_interleave(32, dim=0)
            y = base + torch.randn((len(idx), d), generator=g) * 1e-7
            x[idx] = y / y.norm(dim=1, keepdim=True)
            if c0 == 0:
                corpus_chunks.dup_rows = x[starts[:64] + 5].clone()
            yield c0, x


def make_queries(kind, nq, d):
    g = torch.Generator()
    g.manual_seed(99)
    if kind == "clustered":
        cq = corpus_chunks.centres[torch.randint(0, 4096, (nq,), generator=g)]
        nz = torch.randn((nq, d), generator=g)
        q = cq + 0.3 * nz / nz.norm(dim=1, keepdim=True)
    else:
        q = torch.randn((nq, d), generator=g)
        if kind == "near_duplicate":
            nz = torch.randn((nq // 2, d), generator=g)
            q[0::2] = corpus_chunks.dup_rows[: nq // 2] + 0.02 * nz / nz.norm(dim=1, keepdim=True)
    return q / q.norm(dim=1, keepdim=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    d = 384
    torch.set_num_threads(8)
    names = ["bf16", "i8_index", "i8_tile", "i8_row"]
    for kind in ("isotropic", "clustered", "near_duplicate"):
        t0 = time.time()
        # pass 1: the index-wide scale
        gmax = 0.0
        first = None
        for c0, x in corpus_chunks(kind, n, d):
            gmax = max(gmax, float(x.abs().max()))
            if c0 == 0:
                first = x
        q = make_queries(kind, nq, d)
        qn = q.norm(dim=1)
        # the query images: bf16, and i8 with the query's own scale
        qimg = {"bf16": bf16_round(q)}
        sq = q.abs().max(dim=1, keepdim=True).values / 127.0
        for nm in names[1:]:
            qimg[nm] = quant_i8(q, sq)
        eq = {nm: (q - qimg[nm]).double().norm(dim=1).float() for nm in names}
        v = {nm: torch.empty((n, nq)) for nm in names}
        ex = {nm: torch.empty(n) for nm in names}
        xn = torch.empty(n)
        for c0, x in corpus_chunks(kind, n, d):
            m = x.shape[0]
            xn[c0 : c0 + m] = x.norm(dim=1)
            img = {"bf16": bf16_round(x), "i8_index": quant_i8(x, gmax / 127.0)}
            rmax = x.abs().max(dim=1, keepdim=True).values
            img["i8_row"] = quant_i8(x, rmax / 127.0)
            tmax = rmax.reshape(-1, 32).max(dim=1, keepdim=True).values.repeat_interleave(32, dim=0) if m % 32 == 0 else rmax
            img["i8_tile"] = quant_i8(x, tmax / 127.0)
            for nm in names:
                v[nm][c0 : c0 + m] = img[nm] @ qimg[nm].T
                ex[nm][c0 : c0 + m] = (x - img[nm]).norm(dim=1)
        print(f"\n### {kind}: {n} x {d} unit rows, {nq} queries, k = {k}  (largest |x_i| {gmax:.3f}; {time.time() - t0:.0f} s)")
        print("| filter | max residual norm of a row | mean | query residual | margin (ip units) | listed, launch 1 | launch 2 | evaluated in float64 |")
        print("|---|---|---|---|---|---|---|---|")
        n1 = n // 16
        for nm in names:
            for per_row in (False, True):
                e_max = float(ex[nm].max())
                xmax = float(xn.max())
                l1 = l2 = ev = 0.0
                mgs = []
                for j in range(nq):
                    vj = v[nm][:, j]
                    e_q = float(eq[nm][j])
                    if per_row:
                        mg = ex[nm] * float(qn[j]) + (xn + ex[nm]) * e_q  # each row's own bound
                    else:
                        mg = torch.full((1,), e_max * float(qn[j]) + (xmax + e_max) * e_q).expand(n)
                    mgs.append(float(mg.mean()))
                    lb = vj - mg
                    t0_ = torch.topk(lb[:32768], k).values[-1]          # the sample's threshold
                    c1 = vj[:n1] + mg[:n1] >= t0_                       # passes v >= T0 - mg
                    t1 = torch.topk(lb[:n1][c1], k).values[-1]
                    c2 = vj[n1:] + mg[n1:] >= t1
                    listed = torch.cat([lb[:n1][c1], lb[n1:][c2]])
                    ub = listed + 2 * torch.cat([mg[:n1][c1], mg[n1:][c2]])
                    kv = torch.topk(listed, k).values[-1]
                    l1 += float(c1.sum()); l2 += float(c2.sum()); ev += float((ub >= kv).sum())
                tag = nm + ("_rowm" if per_row else "")
                print(f"| {tag} | {e_max:.5f} | {float(ex[nm].mean()):.5f} | {float(eq[nm].mean()):.5f} | {np.mean(mgs):.5f} | "
                      f"{l1 / nq:.0f} | {l2 / nq:.0f} | {ev / nq:.0f} |", flush=True)
        del v, ex


if __name__ == "__main__":
    main()
