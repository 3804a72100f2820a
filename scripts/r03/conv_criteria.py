"""Convergence step of the predicted covariance under candidate steady-state tests (numpy, batched over draws).
  a: max |dP| <= tol * max |P|                          (round 2: one-step change, checked every 8 steps)
  b: |dP_ij| <= tol * sqrt(P_ii P_jj) for every entry   (entrywise scaled), on TWO consecutive checks
Also: the error of the frozen covariance against the recursion continued for 3000 more steps, entrywise scaled."""
import sys
import numpy as np
sys.path.insert(0, ".")
import workloads as W


def riccati(P, T, Z, H, RQR):
    M = np.einsum("bij,bj->bi", P, Z)
    F = np.einsum("bi,bi->b", Z, M) + H
    Pf = P - M[:, :, None] * M[:, None, :] / F[:, None, None]
    S = T @ Pf @ T.transpose(0, 2, 1) + RQR
    return 0.5 * (S + S.transpose(0, 2, 1))


def run(name, B, n, tols=(1e-15, 2e-15, 4e-15)):
    w = W.WORKLOADS[name](B=B, n=n)
    T, Z, H, R, Q = w["T"], w["Z"][:, 0, :], w["H"][:, 0, 0], w["R"], w["Q"]
    RQR = R @ Q @ R.transpose(0, 2, 1)
    P = 0.5 * (w["P0"] + w["P0"].transpose(0, 2, 1))
    conv = {("a", t): np.full(B, n) for t in tols}
    conv.update({("b", t): np.full(B, n) for t in tols})
    passed = {t: np.zeros(B, bool) for t in tols}
    frozen = {}
    for t in range(n):
        Pn = riccati(P, T, Z, H, RQR)
        if (t & 7) == 7 and t + 16 < n:
            dP = np.abs(Pn - P)
            dg = np.sqrt(np.abs(np.einsum("bii->bi", Pn)))
            sc = dg[:, :, None] * dg[:, None, :]
            for tol in tols:
                oka = dP.max((1, 2)) <= tol * np.abs(Pn).max((1, 2))
                k = ("a", tol)
                new = oka & (conv[k] == n)
                conv[k][new] = t + 1
                okb = (dP <= tol * sc).all((1, 2))
                k = ("b", tol)
                new = okb & passed[tol] & (conv[k] == n)
                conv[k][new] = t + 1
                if tol == tols[0]:
                    for i in np.nonzero(new)[0]:
                        frozen[i] = Pn[i].copy()
                passed[tol] = okb
        P = Pn
    for _ in range(3000):
        P = riccati(P, T, Z, H, RQR)
    print(f"== {name} B={B} n={n}")
    for k, v in sorted(conv.items()):
        grp = v[: (B // 4) * 4].reshape(-1, 4).mean(1)
        print(f"  {k[0]} tol={k[1]:.0e}: median {np.median(v):.0f} mean {v.mean():.1f} p90 {np.percentile(v,90):.0f} "
              f"p99 {np.percentile(v,99):.0f} never {(v == n).sum()} | groups of 4: max {grp.max():.1f}; "
              f"cumulative unconverged at 128k: {[int((v > 128*k_).sum()) for k_ in range(1, 8)]}")
    if frozen:
        errs = []
        for i, Pf in frozen.items():
            dg = np.sqrt(np.abs(np.diag(P[i])))
            sc = np.outer(dg, dg)
            errs.append((np.abs(Pf - P[i]) / np.where(sc > 0, sc, 1)).max())
        errs = np.array(errs)
        print(f"  b tol={tols[0]:.0e}: entrywise-scaled error of the frozen P vs 3000 more steps: max {errs.max():.1e} median {np.median(errs):.1e}")


if __name__ == "__main__":
    run("headline", 4096, 1000)
    run("c3", 2048, 2000)
