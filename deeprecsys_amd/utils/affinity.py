"""Bind an engine process (one per GPU) to the host cores next to its GPU.

The reference starts its engines as plain `multiprocessing.Process`es and leaves their placement
to the scheduler (DeepRecSys.py:62-72): fine for 32 CPU engines, not for one process per GPU on a
two-socket host -- each accelerator engine spins in drs_wait, runs a conversion pool and owns
pinned buffers the GPU reads over PCIe, all of which want the socket the GPU hangs off.  This
module computes, for local rank r of n (rank r drives HIP device r), the cores of that GPU's NUMA
node -- shared evenly among the ranks whose GPUs sit on the same node -- and applies the mask
BEFORE the process allocates pinned memory or starts worker threads.  Where sysfs says nothing
(containers, single-node hosts reporting -1) the process's allowed cores are dealt evenly.

Pure planning (`plan`) is separate from the sysfs readers and from `sched_setaffinity`, so the
masks for 1 / 2 / 4 / 8 ranks are tested on CPU with synthetic topologies.
"""
import glob
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0,1,2,3,8,10,11} (the kernel's cpulist format)."""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.update(range(int(lo), int(hi) + 1))
        else:
            cpus.add(int(part))
    return cpus


def format_cpulist(cpus):
    cpus = sorted(cpus)
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def _even_slice(cpus, k, n):
    """share k of n of a sorted core list: contiguous, sizes differing by at most one, never empty
    while there are at least n cores (with fewer the ranks share)."""
    cpus = sorted(cpus)
    if not cpus:
        return []
    if len(cpus) < n:
        return [cpus[k % len(cpus)]]
    lo = (len(cpus) * k) // n
    hi = (len(cpus) * (k + 1)) // n
    return cpus[lo:hi]


def plan(gpu_nodes, node_cpus, allowed, n_ranks):
    """Core masks for ranks 0 .. n_ranks-1, rank r on GPU r.

    gpu_nodes: NUMA node of each GPU (-1 / None = unknown); node_cpus: {node: set of cores};
    allowed: the cores this process tree may use.  Returns [(cores, source)] with source
    "numa" (the GPU's node, shared evenly among the ranks on that node) or "even" (an even
    deal of the allowed cores: unknown node, or a node with no allowed core)."""
    allowed = set(allowed)
    out = [None] * n_ranks
    by_node = {}
    for r in range(n_ranks):
        node = gpu_nodes[r] if r < len(gpu_nodes) else None
        local = set(node_cpus.get(node, ())) & allowed if node is not None and node >= 0 else set()
        if local:
            by_node.setdefault(node, []).append(r)
        else:
            out[r] = (_even_slice(allowed, r, n_ranks), "even")
    for node, ranks in by_node.items():
        local = set(node_cpus[node]) & allowed
        for k, r in enumerate(ranks):
            out[r] = (_even_slice(local, k, len(ranks)), "numa")
    return out


# ---- sysfs readers -------------------------------------------------------------------------------
def kfd_gpu_pci_addresses(root="/sys/class/kfd/kfd/topology/nodes"):
    """PCI addresses of the GPUs in KFD topology order -- the order HIP numbers its devices in
    (CPU nodes have simd_count 0 and are skipped).  [] when the topology is not readable."""
    addrs = []
    try:
        nodes = sorted((int(os.path.basename(p)), p) for p in glob.glob(os.path.join(root, "*")) if os.path.basename(p).isdigit())
    except OSError:
        return addrs
    for _, p in nodes:
        props = {}
        try:
            with open(os.path.join(p, "properties")) as f:
                for line in f:
                    kv = line.split()
                    if len(kv) == 2:
                        props[kv[0]] = kv[1]
        except OSError:
            continue
        if int(props.get("simd_count", "0")) <= 0:
            continue
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        addrs.append("%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7))
    return addrs


def visible_devices(n_physical):
    """HIP device index -> physical index, honouring ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES
    when they are plain integer lists (UUID forms: identity)."""
    order = list(range(n_physical))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var, "").strip()
        if not v:
            continue
        try:
            pick = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            continue
        order = [order[i] for i in pick if 0 <= i < len(order)]
    return order


def gpu_numa_nodes(pci_root="/sys/bus/pci/devices", kfd_root="/sys/class/kfd/kfd/topology/nodes"):
    """([node of HIP device i], {node: cores}) from sysfs; ([], {}) when nothing can be read."""
    addrs = kfd_gpu_pci_addresses(kfd_root)
    addrs = [addrs[i] for i in visible_devices(len(addrs))]
    nodes, node_cpus = [], {}
    for a in addrs:
        node = -1
        try:
            node = int(open(os.path.join(pci_root, a, "numa_node")).read().strip())
        except (OSError, ValueError):
            pass
        nodes.append(node)
        if node >= 0 and node not in node_cpus:
            try:
                node_cpus[node] = parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read())
            except OSError:
                try:
                    node_cpus[node] = parse_cpulist(open(os.path.join(pci_root, a, "local_cpulist")).read())
                except OSError:
                    pass
    return nodes, node_cpus


def bind_rank(local_rank, local_world, apply=True):
    """Bind this process (local rank r of n, GPU r) and return what was decided, for the run's
    config record: {"cpus": "8-15", "n_cpus": 8, "source": "numa" | "even" | "unchanged", "numa_node": k}.
    DRS_NO_AFFINITY=1 leaves the mask alone."""
    allowed = os.sched_getaffinity(0)
    info = {"cpus": format_cpulist(allowed), "n_cpus": len(allowed), "source": "unchanged", "numa_node": None}
    if os.environ.get("DRS_NO_AFFINITY", "") == "1" or local_world < 1 or not (0 <= local_rank < local_world):
        return info
    nodes, node_cpus = gpu_numa_nodes()
    cores, source = plan(nodes, node_cpus, allowed, local_world)[local_rank]
    if not cores:
        return info
    if local_world == 1 and source == "even":
        return info                           # one rank, no topology: everything it already has
    if apply:
        try:
            os.sched_setaffinity(0, cores)
        except OSError:
            return info
    info.update({"cpus": format_cpulist(cores), "n_cpus": len(cores), "source": source,
                 "numa_node": nodes[local_rank] if local_rank < len(nodes) else None})
    return info
