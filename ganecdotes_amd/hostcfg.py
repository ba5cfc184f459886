"""Host-side settings of the training classes.

The host only feeds the GPU: a latent, a few (2, 512) normals, an angle, a flip and — with pick_source='host' — a randperm per
iteration.  torch's intra-op thread pool defaults to every logical CPU of the machine; inside a container whose cgroup grants a
fraction of them (16 of 256 on the MI355X boxes) each of those tiny CPU ops then wakes a pool it may not run: torch.randperm(65536)
took 25 ms instead of 0.2 ms and the hfc_with_simclr configuration ran at 41 images/s instead of 422 (tools/simclr_check.py).
limit_host_threads() caps the pool at the CPUs this process may actually use — it only ever lowers the count, and it says so
once on the package logger because the setting is PROCESS-WIDE (an embedding application's own CPU work runs on the same
pool).  A process that may use all its CPUs is left alone; GX_HOST_THREADS=n asks for a lower cap."""
import logging
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


def limit_host_threads(cap=None):
    """torch.set_num_threads(min(current, usable CPUs[, cap])), once per process.  cap: an explicit upper bound (argument or
    GX_HOST_THREADS); without one only the CPUs the process may not use anyway are taken away."""
    global _DONE
    if _DONE:
        return torch.get_num_threads()
    _DONE = True
    try:
        cap = int(os.environ["GX_HOST_THREADS"]) if "GX_HOST_THREADS" in os.environ else cap
    except ValueError:
        pass
    before = torch.get_num_threads()
    want = max(1, min(before, usable_cpus(), cap if cap else before))
    if want < before:
        torch.set_num_threads(want)
        logging.getLogger("ganecdotes_amd").warning(
            "torch intra-op threads %d -> %d: the process may use %d CPUs (affinity / cgroup quota)%s; process-wide, "
            "see ganecdotes_amd/hostcfg.py", before, want, usable_cpus(), f", cap {cap}" if cap else "")
    return torch.get_num_threads()
