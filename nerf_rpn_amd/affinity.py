"""Rank -> host-core pinning for the one-process-per-GPU launchers (bench.py, run_rpn.py / run_fcos.py main_worker, run_rpn_detect.py).

A training step is 200-900 C-ABI launches from Python (bench.py ``host``): with eight such processes on one host the enqueue threads must not
migrate between sockets or share a core.  Each rank gets a contiguous share of the cores of the NUMA node its GPU hangs off (sysfs:
/sys/bus/pci/devices/<bdf>/numa_node, /sys/devices/system/node/node<N>/cpulist), divided among ALL the GPUs of that node -- a rank takes
the share of its GPU's place on the host, so independent jobs on one node do not pile onto the same cores; when the topology cannot be read
(containers without sysfs, one node) the allowed cores are simply divided evenly among this job's ranks.  NRPN_PIN=0 switches it off.
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


def _gpu_bdf(device_index):
    import torch
    p = torch.cuda.get_device_properties(device_index)
    return f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"


def _node_gpu_slot(device_index):
    """(position of this GPU among ALL AMD accelerators attached to its NUMA node, how many there are) from sysfs, or None.  The share of a
    rank is cut by the GPU's place on the HOST, not by its rank inside this job: two jobs on one node (GPUs 0-3 and 4-7, or several single-GPU
    launches) then land on disjoint cores instead of both starting at the first core of the node."""
    try:
        mine = _gpu_bdf(device_index)
        node = _gpu_numa_node(device_index)
        if node is None:
            return None
        gpus = []
        for bdf in sorted(os.listdir("/sys/bus/pci/devices")):
            base = f"/sys/bus/pci/devices/{bdf}"
            try:
                with open(base + "/vendor") as f:
                    if f.read().strip() != "0x1002":
                        continue
                with open(base + "/class") as f:
                    cls = f.read().strip()
                if not (cls.startswith("0x12") or cls.startswith("0x0302") or cls.startswith("0x0380")):    # processing accelerator / 3D / display controller
                    continue
                with open(base + "/numa_node") as f:
                    if int(f.read().strip()) != node:
                        continue
            except (OSError, ValueError):
                continue
            gpus.append(bdf)
        return (gpus.index(mine), len(gpus)) if mine in gpus else None
    except Exception:
        return None


def plan(local_rank, local_world, device_indices=None, allowed=None, numa_of=None, cpus_of=None, slot_of=None):
    """-> (cores for this rank, numa node or None).  Pure function of its arguments when ``numa_of`` / ``cpus_of`` / ``slot_of`` are given
    (tests).  ``slot_of(device) -> (slot, gpus on the node) | None``: with it the node's cores are divided among all GPUs of the node and the
    rank takes the share of ITS GPU; without it (no sysfs) among this job's ranks on the node."""
    allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    device_indices = list(device_indices) if device_indices is not None else list(range(local_world))
    numa_of = numa_of or _gpu_numa_node
    cpus_of = cpus_of or _node_cpus
    if slot_of is None and numa_of is _gpu_numa_node:
        slot_of = _node_gpu_slot
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
    k, parts = peers.index(local_rank), len(peers)
    if mine is not None and slot_of is not None:
        slots = [slot_of(device_indices[r]) for r in peers]
        if all(sl is not None for sl in slots) and len({sl[1] for sl in slots}) == 1 and slots[0][1] >= len(peers) and len(pool) >= slots[0][1]:
            k, parts = slots[peers.index(local_rank)][0], slots[0][1]       # the GPU's place among the node's GPUs
    per = max(1, len(pool) // parts)
    cores = pool[k * per:(k + 1) * per] if k * per < len(pool) else [pool[k % len(pool)]]
    return cores, mine


def pin_rank(local_rank, local_world, device_indices=None):
    """Pin the calling process (all of its current threads inherit nothing: call before the heavy imports spawn pools where possible) and
    cap torch's intra-op CPU threads to the share.  -> a dict for logs / bench.py's JSON line."""
    info = {"host_cores": os.cpu_count(), "allowed_cores": len(os.sched_getaffinity(0)), "pinned": False}
    if os.environ.get("NRPN_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity") or local_world <= 1:
        return info         # a single rank shares the host with nobody (and bench.py's cpu_baseline leg wants every core)
    if info["allowed_cores"] < (info["host_cores"] or 0):
        # somebody (a job scheduler, taskset, a container) already restricted this process: the shares below are cut from THAT mask; said once
        info["note"] = f"affinity mask already restricted to {info['allowed_cores']} of {info['host_cores']} cores: shares are cut from it"
        if local_rank == 0 and os.environ.get("NRPN_QUIET") != "1":
            import sys
            print(f"[nerf_rpn_amd] {info['note']} (NRPN_PIN=0 leaves placement to the launcher)", file=sys.stderr, flush=True)
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
