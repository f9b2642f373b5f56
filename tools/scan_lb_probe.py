"""Phase stamps of the look-back forward (a -DLB_PROBE build: APERTIS_HIP_LIB=apertis_llm_amd/libapertis_hip_probe.so).
    python tools/scan_lb_probe.py [B L h]"""
import ctypes, math, sys, torch
sys.path.insert(0, ".")
from apertis_llm_amd import ops, _lib
dev = torch.device("cuda:0")
B, L, h = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (44, 4096, 11)
N = 16
Dn, R = h * N, math.ceil(h * 64 / 16)
Wb, Wr = -(-Dn // 64) * 64, -(-R // 64) * 64
p = torch.randn(B, L, 2 * Wb + Wr, device=dev).bfloat16()
xz = torch.randn(B, L, 2 * Dn, device=dev).bfloat16()
xc = torch.randn(B, L, Dn, device=dev).bfloat16()
dl = torch.randn(B, L, h, device=dev) - 4
A = torch.empty(h, N, device=dev).uniform_(math.log(.5), math.log(.99))
D = torch.ones(Dn, device=dev)
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
g = Dn // 4
Rr = 64 // g
nitems = -(-L // 64) * -(-B // Rr)
probe = torch.zeros(nitems, 2, 8, dtype=torch.int64, device=dev)
flush = torch.empty(1 << 28, dtype=torch.int32, device=dev)


def fwd():
    Btp, Cp, _ = ops.split_cols(p, (Wb, Wb, Wr))
    _, z = ops.split_cols(xz, (Dn, Dn))
    with torch.no_grad():
        return ops.scan_gate(dl, A, Btp, Cp, xc, z, D, delta_softplus=True)


for _ in range(3):
    fwd()
flush.sum()
raw.apertis_scan_lookback_set_probe(ctypes.c_void_p(probe.data_ptr()))
fwd()
torch.cuda.synchronize()
raw.apertis_scan_lookback_set_probe(ctypes.c_void_p(0))
t = probe.cpu().double() * 0.01          # us (100 MHz)
t0 = t[:, :, 0].min()
names = ["entry", "ticket", "loads issued", "aggregate done", "barrier 1", "publish + poll done", "barrier 2", "end"]
print(f"B={B} L={L} Dn={Dn}: {nitems} work-groups; kernel span {float(t[:, :, 7].max() - t0):.1f} us")
for w, nm in ((0, "wave 0"), (1, "wave 3")):
    d = t[:, w, 1:] - t[:, w, :-1]
    print(f"  {nm}: mean phase durations (us) " + "  ".join(f"{names[k + 1]} {float(d[:, k].mean()):.2f}" for k in range(7)) +
          f"   | life {float((t[:, w, 7] - t[:, w, 0]).mean()):.2f}")
    q = torch.quantile(d, torch.tensor([0.5, 0.9, 0.99], dtype=torch.float64), dim=0)
    for nmq, row in zip(("p50", "p90", "p99"), q):
        print(f"     {nmq}: " + "  ".join(f"{float(x):6.2f}" for x in row))
# by ticket order: start time of items, in deciles
st = t[:, 0, 0] - t0
idx = torch.linspace(0, nitems - 1, 11).long()
print("  start time of ticket deciles (us):", " ".join(f"{float(st[i]):.1f}" for i in idx))
en = t[:, 0, 7] - t0
print("  end   time of ticket deciles (us):", " ".join(f"{float(en[i]):.1f}" for i in idx))
