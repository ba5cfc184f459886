"""finish (HBM-bound) and the finest TN GEMM (MFMA-bound), sequential vs pipelined by view on two streams"""
import sys, torch
sys.path.insert(0, ".")
from ganecdotes_amd import ops
dev = torch.device("cuda")
B2, H, D, K = 32, 256, 512, 8
torch.manual_seed(0)
Z = torch.randn(B2, H, H, D, device=dev)
C = ops.lowrank_coeffs(B2, H * H, K, dev); C.normal_()
Wc = torch.randn(K, D, device=dev); gWc = torch.zeros(K, D, device=dev)
ta = torch.randn(B2 * H * H, 128, device=dev); tb = torch.randn(B2 * H * H, 128, device=dev)
gw = torch.zeros(D, 256, device=dev)
rows = B2 * H * H
def split_for(r):
    tiles = 4 * 2
    return int(max(1, min(512 // tiles, r // 256)))
def seq():
    G, Gp = ops.proto_lowrank_finish(C, Z, Wc, gWc, 0.0, want_pool=True)
    ops.gemm_tn_cat(G.reshape(rows, D), ta, tb, gw, beta=0.0, split_k=split_for(rows))
    return G, Gp
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def piped():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    hr = rows // 2
    keep = []
    for h in range(2):
        with torch.cuda.stream(s1):
            G, Gp = ops.proto_lowrank_finish(C[h * hr:(h + 1) * hr], Z[h * 16:(h + 1) * 16], Wc, gWc, float(h), want_pool=True)
            ev = torch.cuda.Event(); ev.record(s1)
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            ops.gemm_tn_cat(G.reshape(hr, D), ta[h * hr:(h + 1) * hr], tb[h * hr:(h + 1) * hr], gw, beta=float(h), split_k=split_for(hr))
        keep.append((G, Gp))
    cur.wait_stream(s1); cur.wait_stream(s2)
    return keep
def timeit(fn, n=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): r = fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
seq(); ref = gw.clone()
piped(); d = (gw - ref).abs().max().item() / ref.abs().max().item()
for _ in range(2):
    print(f"sequential {timeit(seq):.3f} ms | pipelined by view {timeit(piped):.3f} ms   (rel diff of dWp {d:.2e})", flush=True)
