"""Does a hipGraph over the generator forward of the bench batch (32 images, training views: no RGB) save the launch gaps?
Eager against torch.cuda.CUDAGraph replay of the same launches, interleaved."""
import sys
import torch
sys.path.insert(0, ".")
import bench

dev = torch.device("cuda", 0)
g = bench.build_generator(256, dev)
torch.manual_seed(0)
lat = torch.randn(32, g.n_latent, 512, device=dev)
ml = g.mean_latent(128)


g.want_image = False


def fwd():
    return g([lat], truncation=0.7, truncation_latent=ml, input_is_latent=True, randomize_noise=False)


for _ in range(3):
    out = fwd()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        fwd()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(graph):
    out_g = fwd()
torch.cuda.synchronize()


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for _ in range(3):
    print(f"eager {timeit(fwd):7.3f} ms   graph replay {timeit(graph.replay):7.3f} ms", flush=True)
ref = fwd()
graph.replay()
torch.cuda.synchronize()
print("taps equal:", all(torch.equal(a, b) for a, b in zip(ref[1], out_g[1])))
