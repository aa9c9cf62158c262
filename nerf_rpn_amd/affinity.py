"""Rank -> host-core pinning for the one-process-per-GPU launchers (bench.py, run_rpn.py / run_fcos.py main_worker, run_rpn_detect.py).

A training step is 200-900 C-ABI launches from Python (bench.py ``host``): with eight such processes on one host the enqueue threads must not
migrate between sockets or share a core.  Each rank gets a contiguous share of the cores of the NUMA node its GPU hangs off (sysfs:
/sys/bus/pci/devices/<bdf>/numa_node, /sys/devices/system/node/node<N>/cpulist), divided among the local ranks on that node; when the
topology cannot be read (containers without sysfs, one node) the allowed cores are simply divided evenly.  NRPN_PIN=0 switches it off.
The reference leaves placement to the OS (run_rpn.py:620-691 spawns its workers unpinned)."""
import os


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def _gpu_numa_node(device_index):
    """NUMA node of a visible HIP device, or None."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except Exception:
        return None


def _node_cpus(node):
    try:
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            return _parse_cpulist(f.read())
    except Exception:
        return None


def plan(local_rank, local_world, device_indices=None, allowed=None, numa_of=None, cpus_of=None):
    """-> (cores for this rank, numa node or None).  Pure function of its arguments when ``numa_of`` / ``cpus_of`` are given (tests)."""
    allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    device_indices = list(device_indices) if device_indices is not None else list(range(local_world))
    numa_of = numa_of or _gpu_numa_node
    cpus_of = cpus_of or _node_cpus
    nodes = [numa_of(d) for d in device_indices]
    mine = nodes[local_rank]
    pool, peers = None, None
    if mine is not None and all(n is not None for n in nodes):
        cpus = cpus_of(mine)
        if cpus:
            pool = [c for c in sorted(cpus) if c in set(allowed)]
            peers = [r for r, n in enumerate(nodes) if n == mine]
    if not pool or len(pool) < len(peers or [0]):
        pool, peers, mine = allowed, list(range(local_world)), None
    k = peers.index(local_rank)
    per = max(1, len(pool) // len(peers))
    cores = pool[k * per:(k + 1) * per] if k * per < len(pool) else [pool[k % len(pool)]]
    return cores, mine


def pin_rank(local_rank, local_world, device_indices=None):
    """Pin the calling process (all of its current threads inherit nothing: call before the heavy imports spawn pools where possible) and
    cap torch's intra-op CPU threads to the share.  -> a dict for logs / bench.py's JSON line."""
    info = {"host_cores": os.cpu_count(), "allowed_cores": len(os.sched_getaffinity(0)), "pinned": False}
    if os.environ.get("NRPN_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity") or local_world <= 1:
        return info         # a single rank shares the host with nobody (and bench.py's cpu_baseline leg wants every core)
    try:
        cores, node = plan(local_rank, local_world, device_indices)
        os.sched_setaffinity(0, cores)        # the calling thread; threads it creates later (autograd's device thread, the loader) inherit
        try:                                  # threads that already exist (torch's pools started at import) follow
            for tid in os.listdir("/proc/self/task"):
                try:
                    os.sched_setaffinity(int(tid), cores)
                except OSError:
                    pass
        except OSError:
            pass
        import torch
        torch.set_num_threads(max(1, min(len(cores), 16)))
        info.update({"pinned": True, "cores": [cores[0], cores[-1]], "n_cores": len(cores), "numa_node": node})
    except Exception as e:                  # placement is an optimisation: never fail a run over it
        info["error"] = str(e)[:200]
    return info
