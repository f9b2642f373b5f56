"""CPU baseline leg of bench.py — ORACLE CODE, test/measurement infrastructure only.

Times the CPU restatement (oracle/ref_cpu.py, verified against the reference in the build
container) on the GPU box's host cores for a BOUNDED sample of the benchmark workload and
extrapolates to the whole model.  It is a reported baseline ("port"), never the target.
"""
import math
import os
import time

import torch
import torch.nn.functional as F

from . import ref_cpu


def _random_layer_state(H, h, N, I, E, R, moe, gen):
    Dn = h * N
    r = lambda *s: torch.randn(*s, generator=gen) * 0.02
    sd = {}
    p = "model.layers.0.attention."
    sd[p + "pre_norm.weight"], sd[p + "pre_norm.bias"] = torch.ones(H), torch.zeros(H)
    q = p + "attention_mechanism_impl."
    sd[q + "in_proj_x.weight"], sd[q + "in_proj_z.weight"] = r(Dn, H), r(Dn, H)
    sd[q + "conv1d.weight"], sd[q + "conv1d.bias"] = torch.randn(Dn, 1, 4, generator=gen) * 0.5, torch.zeros(Dn)
    sd[q + "x_param_proj.weight"] = r(R + 2 * Dn, Dn)
    sd[q + "dt_proj_head.weight"] = r(h, R)
    sd[q + "dt_proj_head.bias"] = torch.empty(h).uniform_(math.log(1e-3), math.log(1e-2), generator=gen)
    sd[q + "A_log"] = torch.empty(h, N).uniform_(math.log(0.5), math.log(0.99), generator=gen)
    sd[q + "D"] = torch.ones(Dn)
    sd[q + "out_proj.weight"] = r(H, Dn)
    f = "model.layers.0.feed_forward."
    sd[f + "pre_norm.weight"], sd[f + "pre_norm.bias"] = torch.ones(H), torch.zeros(H)
    if moe:
        g = f + "ffn."
        sd[g + "router_norm.weight"], sd[g + "router_norm.bias"] = torch.ones(H), torch.zeros(H)
        sd[g + "router.weight"], sd[g + "router.bias"] = r(E, H), torch.zeros(E)
        for e in range(E):
            sd[f"{g}experts.{e}.0.weight"], sd[f"{g}experts.{e}.0.bias"] = torch.ones(H), torch.zeros(H)
            sd[f"{g}experts.{e}.1.weight"], sd[f"{g}experts.{e}.1.bias"] = r(I, H), torch.zeros(I)
            sd[f"{g}experts.{e}.4.weight"], sd[f"{g}experts.{e}.4.bias"] = r(H, I), torch.zeros(H)
    else:
        sd[f + "ffn.0.weight"], sd[f + "ffn.0.bias"] = r(I, H), torch.zeros(I)
        sd[f + "ffn.3.weight"], sd[f + "ffn.3.bias"] = r(H, I), torch.zeros(H)
    return sd


def cpu_model():
    """The host CPU's model string ("state the core count and CPU model")."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def usable_cpus():
    """(os.cpu_count(), CPUs this process may actually use): the scheduler affinity and the cgroup's CPU quota both bound it.
    On the GPU boxes os.cpu_count() reports the whole host (256) while the container's share is 16: a thread pool of 256 on
    16 CPUs' worth of quota does not finish the sample (measured: > 360 s against 15 s)."""
    total = os.cpu_count() or 1
    use = total
    try:
        use = min(use, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = parts[0], float(parts[1])
            else:
                quota = parts[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                    period = float(g.read().split()[0])
            if quota not in ("max", "-1"):
                use = min(use, max(1, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return total, max(1, use)


def _time_scan_only(h, N, L, threads, gen):
    """Forward of the two scan restatements alone at B=1 (SURVEY section 8(d) tensors), effective GB/s on the GPU's formula
    T*(3*Dn*e + 4*h) with e = 4 (this oracle runs fp32)."""
    torch.set_num_threads(threads)
    Dn = h * N
    delta = F.softplus(torch.randn(1, L, h, generator=gen) - 4.0)
    A_log = torch.empty(h, N).uniform_(math.log(0.5), math.log(0.99), generator=gen)
    Bt, C = torch.randn(1, L, Dn, generator=gen), torch.randn(1, L, Dn, generator=gen)
    nbytes = L * (3 * Dn * 4 + 4 * h)
    out = {}
    with torch.no_grad():
        for name, fn in (("recurrent", ref_cpu.scan_recurrent), ("vectorised", ref_cpu.scan_chunked_vectorised)):
            fn(delta, A_log, Bt, C)
            t0 = time.perf_counter()
            fn(delta, A_log, Bt, C)
            out[name] = nbytes / (time.perf_counter() - t0) / 1e9
    return out


def time_layer(H, h, N, I, E, K, moe, L, vocab, n_layers_total, threads=None, reps=2, emit=None):
    """fwd+bwd of ONE layer of the given shape at B=1 and the benchmark's sequence length L, in TRAIN mode (expert
    capacity on, dropout p = 0 for determinism), plus the lm_head + CE, through the oracle with torch autograd:
    one warm-up repetition, then the median of `reps`; extrapolated to n_layers_total layers.  Three legs (BASELINE.md
    section 3): the sequential-recurrence scan (core.py:337-353, what the reference trainer executes) on every host core
    this process may use (`value` = `value_recurrent`; `cores` = min(os.cpu_count(), affinity, cgroup quota): the GPU boxes
    report 256 host cores to a container that owns 16), the same on 32 threads (`value_recurrent_32t`, round 5's figure), and the vectorised scan (core.py:324-335 per 64-token chunk, `value_vectorised`, the
    fair CPU comparison).  Returns a dict."""
    avail, usable = usable_cpus()
    threads = threads or usable          # every core this process may use, not the host's count
    gen = torch.Generator().manual_seed(0)
    R = math.ceil(H / 16)
    sd = _random_layer_state(H, h, N, I, E, R, moe, gen)
    for v in sd.values():
        v.requires_grad_(True)
    emb = (torch.randn(vocab, H, generator=gen) * 0.02).requires_grad_(True)
    ids = torch.randint(4, vocab, (1, L), generator=gen)

    def one_layer(scan):
        x = F.embedding(ids, emb)
        lp = "model.layers.0."
        hh = F.layer_norm(x, (H,), sd[lp + "attention.pre_norm.weight"], sd[lp + "attention.pre_norm.bias"], 1e-12)
        x = x + ref_cpu.ssm_layer(sd, lp + "attention.attention_mechanism_impl.", hh, h, N, R, scan=scan)
        hh = F.layer_norm(x, (H,), sd[lp + "feed_forward.pre_norm.weight"], sd[lp + "feed_forward.pre_norm.bias"], 1e-12)
        if moe:
            f, lb, rz, _ = ref_cpu.moe_layer(sd, lp + "feed_forward.ffn.", hh, E, K, "gelu", 1e-12, training=True)
            return x + f, lb + rz
        f = F.linear(F.gelu(F.linear(hh, sd[lp + "feed_forward.ffn.0.weight"], sd[lp + "feed_forward.ffn.0.bias"])),
                     sd[lp + "feed_forward.ffn.3.weight"], sd[lp + "feed_forward.ffn.3.bias"])
        return x + f, x.new_zeros(())

    def one_rep(scan):
        for v in sd.values():
            v.grad = None
        emb.grad = None
        t0 = time.perf_counter()
        y, aux = one_layer(scan)
        t_fwd = time.perf_counter() - t0
        t0 = time.perf_counter()
        (y.sum() + aux).backward()
        t_bwd = time.perf_counter() - t0
        # head: final LN + tied lm_head + CE, fwd+bwd
        xh = y.detach().requires_grad_(True)
        t0 = time.perf_counter()
        logits = F.linear(F.layer_norm(xh, (H,)), emb)
        loss = F.cross_entropy(logits[:, :-1].reshape(-1, vocab), ids[:, 1:].reshape(-1))
        loss.backward()
        t_head = time.perf_counter() - t0
        return t_fwd, t_bwd, t_head

    def leg(scan, nthreads, warm):
        torch.set_num_threads(nthreads)
        if warm:
            one_rep(scan)                                       # warm-up: allocator, thread pool, page-in
        runs = [one_rep(scan) for _ in range(reps)]
        med = lambda i: sorted(r[i] for r in runs)[len(runs) // 2]
        t_fwd, t_bwd, t_head = med(0), med(1), med(2)
        return L / (n_layers_total * (t_fwd + t_bwd) + t_head), (t_fwd, t_bwd, t_head)

    # legs in the order of their cost certainty: <= 32 threads first (round 5's figure: known to finish in ~20 s), its result
    # is emitted at once (`emit`: bench.py keeps the LAST line it could parse, also when its time-out ends this process); the
    # all-usable-cores leg, when it is a different thread count, comes last
    t32 = min(32, threads)
    v_rec, (t_fwd, t_bwd, t_head) = leg(ref_cpu.scan_recurrent, t32, True)
    v_vec, (v_fwd, v_bwd, _) = leg(ref_cpu.scan_chunked_vectorised, t32, True)
    scan_gbps = _time_scan_only(h, N, L, t32, gen)

    def result(v_all):
        best, cores = (v_all, threads) if v_all is not None and v_all > v_rec else (v_rec, t32)
        return {"value": best, "unit": "tokens/s", "cores": cores, "cores_available": avail, "cores_usable": usable,
                "cpu_model": cpu_model(), "kind": "port", "value_recurrent": best, "value_vectorised": v_vec,
                "value_recurrent_32t": v_rec, "threads_second_figure": t32,
                "value_recurrent_all_usable_cores": v_all, "threads_all_usable": threads,
                "scan_fwd_gbps": {"recurrent": scan_gbps["recurrent"], "vectorised": scan_gbps["vectorised"],
                                  "formula": "T*(3*Dn*4 + 4*h) bytes / forward time, B=1, fp32"},
                "sample": f"oracle (torch-CPU restatement) fwd+bwd of 1 of {n_layers_total} layers + lm_head/CE at B=1 L={L}, "
                          f"fp32, train mode (expert capacity on, dropout p=0), 1 warm-up + median of {reps}; at {t32} threads: "
                          f"recurrent scan (as the reference trainer executes it, core.py:337-353) layer fwd {t_fwd:.2f}s bwd "
                          f"{t_bwd:.2f}s head {t_head:.2f}s; vectorised scan (core.py:324-335 per 64-token chunk) layer fwd "
                          f"{v_fwd:.2f}s bwd {v_bwd:.2f}s; step time extrapolated as {n_layers_total}x layer + head; `value` = the "
                          f"faster recurrent figure of the {t32}-thread and the all-usable-cores ({threads}) legs"
                          + ("" if v_all is not None or threads == t32 else " (the latter not finished when this line was written)")}

    if emit is not None:
        emit(result(None))
    v_all = leg(ref_cpu.scan_recurrent, threads, False)[0] if threads != t32 else v_rec
    return result(v_all)


if __name__ == "__main__":   # child process of bench.py: prints one JSON object
    import json
    import sys
    a = [int(v) for v in sys.argv[1:]]
    H, h, N, I, E, K, moe, L, vocab, layers = a
    out = lambda d: print(json.dumps(d), flush=True)
    out(time_layer(H, h, N, I, E, K, bool(moe), L, vocab, layers, emit=out))
