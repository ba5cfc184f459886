"""Host-side settings of the training classes.

The host only feeds the GPU: a latent, a few (2, 512) normals, an angle, a flip and — with pick_source='host' — a randperm per
iteration.  torch's intra-op thread pool defaults to every logical CPU of the machine; inside a container whose cgroup grants a
fraction of them (16 of 256 on the MI355X boxes) each of those tiny CPU ops then wakes a pool it may not run: torch.randperm(65536)
took 25 ms instead of 0.2 ms and the hfc_with_simclr configuration ran at 41 images/s instead of 422 (tools/simclr_check.py).
limit_host_threads() caps the pool at the CPUs this process may actually use — it only ever lowers the count."""
import os

import torch

_DONE = False


def usable_cpus():
    """CPUs this process may use: scheduler affinity, cut by the cgroup CPU quota where there is one"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = float(txt[0])
                if q > 0:
                    n = min(n, max(1, int(q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def limit_host_threads(cap=16):
    """torch.set_num_threads(min(current, usable CPUs, cap)), once per process (GX_HOST_THREADS overrides the cap)"""
    global _DONE
    if _DONE:
        return torch.get_num_threads()
    _DONE = True
    try:
        cap = int(os.environ.get("GX_HOST_THREADS", cap))
    except ValueError:
        pass
    want = max(1, min(torch.get_num_threads(), usable_cpus(), cap))
    if want < torch.get_num_threads():
        torch.set_num_threads(want)
    return torch.get_num_threads()
