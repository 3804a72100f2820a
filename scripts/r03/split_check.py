"""Split forward pipeline (recursion chunks + time-parallel steady kernels) against the step-by-step kernel and the C oracle,
and its timing next to the in-kernel steady phase (KF_B4_SPLIT=0) and KF_FLAG_EXACT_RECURSION."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import workloads as W
from oracle import c_oracle as co
from pymc_statespace_amd import KalmanEngine

eng = KalmanEngine()
KEYS = ("y", "a0", "P0", "T", "Z", "R", "H", "Q")


def ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def check(name, w, ft="standard", oracle_draws=64, full=True, time_it=True, **kw):
    dev = [eng.to_device(w[k]) for k in KEYS]
    fkw = dict(filter_type=ft, mvn_const="exact", store_ll_obs=full, store_filtered=full, store_predicted=full, **kw)
    os.environ["KF_B4_SPLIT"] = "1"
    rs = eng.forward(*dev, **fkw)
    torch.cuda.synchronize()
    rx = eng.forward(*dev, exact_recursion=True, **fkw)
    torch.cuda.synchronize()
    out = [f"{name}: B={w['B']} n={w['n']} m={w['m']}"]
    rel = ((rs.logp - rx.logp).abs() / rx.logp.abs().clamp_min(1.0)).max().item()
    out.append(f"split vs exact: rel dlogp {rel:.2e}")
    if full:
        for nm in ("ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs"):
            a, b = getattr(rs, nm), getattr(rx, nm)
            out.append(f"{nm} {(a - b).abs().max().item():.1e}")
    assert int(rs.info.abs().sum().item()) == int(rx.info.abs().sum().item()), "info differs"
    if oracle_draws:
        k = min(oracle_draws, w["B"])
        ref = co.kalman_filter(w["y"], *[w[x][:k] for x in KEYS[1:]], filter_type=ft, mvn_const="exact", full=full)
        dl = np.abs(rs.logp[:k].cpu().numpy() - ref["logp"])
        out.append(f"vs oracle: max abs dlogp {dl.max():.2e} (|logp| <= {np.abs(ref['logp']).max():.0f})")
        if full:
            out.append(f"pred_P {np.abs(rs.predicted_covs[:k].cpu().numpy() - ref['predicted_covs']).max():.1e} "
                       f"filt_a {np.abs(rs.filtered_states[:k].cpu().numpy() - ref['filtered_states']).max():.1e}")
    print(" | ".join(out), flush=True)
    if time_it:
        t_split = ms(lambda: eng.forward(*dev, out=rs, **fkw))
        os.environ["KF_B4_SPLIT"] = "0"
        t_in = ms(lambda: eng.forward(*dev, out=rs, **fkw))
        os.environ["KF_B4_SPLIT"] = "1"
        t_ex = ms(lambda: eng.forward(*dev, out=rx, exact_recursion=True, **fkw))
        print(f"    time: split {t_split:.3f} ms | in-kernel steady {t_in:.3f} ms | exact {t_ex:.3f} ms", flush=True)
    del dev, rs, rx
    torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "small"):
        for m in (1, 2, 3, 5, 6, 7, 9, 10):
            for n in (256, 300, 1000):
                w = W.random_dense(37, n, 1, m, min(m, 3), seed=100 + m)
                check(f"dense m={m}", w, time_it=False, oracle_draws=37)
        w = W.structural(B=333, n=700)
        y = w["y"].copy()
        y[[5, 130, 131, 400]] = np.nan
        w = dict(w, y=y)
        check("structural + NaN rows", w, time_it=False)
        check("structural logp only", W.structural(B=333, n=700), full=False, time_it=False)
        check("structural cholesky", W.structural(B=100, n=2000), ft="cholesky", time_it=False)
        check("structural univariate", W.structural(B=100, n=600), ft="univariate", time_it=False)
    if which in ("all", "headline"):
        check("headline", W.WORKLOADS["headline"](), oracle_draws=256)
        check("headline logp-only", W.WORKLOADS["headline"](), oracle_draws=256, full=False)
        check("headline B=512", W.WORKLOADS["headline"](B=512), oracle_draws=64)
    if which == "occ":
        for B in (256, 512, 1024, 1536, 2048, 3072, 4096):
            w = W.WORKLOADS["headline"](B=B)
            dev = [eng.to_device(w[k]) for k in KEYS]
            fkw = dict(filter_type="standard", mvn_const="exact")
            rs = eng.forward(*dev, exact_recursion=True, **fkw)
            t_full = ms(lambda: eng.forward(*dev, out=rs, exact_recursion=True, **fkw))
            lo = eng.forward(*dev, exact_recursion=True, store_ll_obs=False, store_filtered=False, store_predicted=False, **fkw)
            t_lo = ms(lambda: eng.forward(*dev, out=lo, exact_recursion=True, store_ll_obs=False, store_filtered=False,
                                          store_predicted=False, **fkw))
            os.environ["KF_FORWARD_B4P"] = "1"
            rp = eng.forward(*dev, exact_recursion=True, **fkw)
            t_pair = ms(lambda: eng.forward(*dev, out=rp, exact_recursion=True, **fkw))
            os.environ["KF_FORWARD_B4P"] = "0"
            dlp = ((rp.logp - rs.logp).abs() / rs.logp.abs().clamp_min(1)).max().item()
            dP = (rp.predicted_covs - rs.predicted_covs).abs().max().item()
            print(f"exact recursion B={B}: full tapes {t_full:.3f} ms, logp only {t_lo:.3f} ms | pair kernel {t_pair:.3f} ms "
                  f"(rel dlogp {dlp:.1e}, dP {dP:.1e})", flush=True)
            del rp
            del dev, rs, lo
    if which == "tune":
        w = W.WORKLOADS["headline"]()
        dev = [eng.to_device(w[k]) for k in KEYS]
        fkw = dict(filter_type="standard", mvn_const="exact")
        rs = eng.forward(*dev, **fkw)
        os.environ["KF_B4_SPLIT"] = "1"
        for ch in (128, 192, 256):
            for wgs in (32, 64, 96, 128, 192, 256, 512, 2048):
                os.environ["KF_B4_SPLIT_CHUNK"] = str(ch)
                os.environ["KF_B4_FILL_WGS"] = str(wgs)
                print(f"chunk {ch} throttled fill workgroups {wgs}: {ms(lambda: eng.forward(*dev, out=rs, **fkw)):.3f} ms", flush=True)
        os.environ.pop("KF_B4_FILL_WGS")
        for ch in (128, 256):
            for dbg in (1, 2, 3):
                os.environ["KF_B4_SPLIT_CHUNK"] = str(ch)
                os.environ["KF_B4_SPLIT_DEBUG"] = str(dbg)
                print(f"chunk {ch} skip(fill=1,scan=2) {dbg}: {ms(lambda: eng.forward(*dev, out=rs, **fkw)):.3f} ms", flush=True)
        os.environ.pop("KF_B4_SPLIT_CHUNK"); os.environ.pop("KF_B4_SPLIT_DEBUG")
    if which in ("all", "c3"):
        check("c3", W.WORKLOADS["c3"](), ft="cholesky", oracle_draws=64)
    if which == "c3tune":
        w = W.WORKLOADS["c3"]()
        dev = [eng.to_device(w[k]) for k in KEYS]
        fkw = dict(filter_type="cholesky", mvn_const="exact")
        rs = eng.forward(*dev, **fkw)
        for split in ("0", "1"):
            os.environ["KF_B4_SPLIT"] = split
            for ch in (256, 384, 512, 768):
                for wgs in (512, 1024, 2048):
                    os.environ["KF_B4_SPLIT_CHUNK"] = str(ch)
                    os.environ["KF_B4_FILL_WGS"] = str(wgs)
                    print(f"c3 split={split} chunk {ch} fill wgs {wgs}: {ms(lambda: eng.forward(*dev, out=rs, **fkw), 5):.3f} ms", flush=True)
                    if split == "0":
                        break
                if split == "0":
                    break
        print(f"c3 exact: {ms(lambda: eng.forward(*dev, out=rs, exact_recursion=True, **fkw), 5):.3f} ms")
