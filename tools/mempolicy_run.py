"""Run a Python script with the task memory policy set first (set_mempolicy(2) through libc's syscall(), before anything creates a
thread: threads inherit it).  An explicit policy takes the process out of the kernel's automatic NUMA balancing, whose periodic
PTE scans invalidate the user-pointer mappings the HSA runtime keeps for host-visible memory; the driver answers each
invalidation by evicting and restoring every queue of the process.
usage: python tools/mempolicy_run.py <mode: none|preferred|bind> <script.py> [args...]"""
import ctypes, os, runpy, sys

def vmstat():
    want = ("numa_pte_updates", "numa_hint_faults", "numa_pages_migrated")
    d = {}
    for ln in open("/proc/vmstat"):
        k, v = ln.split()
        if k in want:
            d[k] = int(v)
    return d

def set_policy(mode):
    cpu = sorted(os.sched_getaffinity(0))[0]
    node = 0
    for n in sorted(os.listdir("/sys/devices/system/node")):
        if n.startswith("node") and os.path.exists("/sys/devices/system/node/%s/cpu%d" % (n, cpu)):
            node = int(n[4:])
    MPOL_PREFERRED, MPOL_BIND = 1, 2
    mask = ctypes.c_ulong(1 << node)
    libc = ctypes.CDLL(None, use_errno=True)
    rc = libc.syscall(238, MPOL_BIND if mode == "bind" else MPOL_PREFERRED, ctypes.byref(mask), 64)      # __NR_set_mempolicy (x86-64)
    return node, rc, ctypes.get_errno()

if __name__ == "__main__":
    mode = sys.argv[1]
    try:
        print("# numa_balancing =", open("/proc/sys/kernel/numa_balancing").read().strip(), file=sys.stderr)
    except OSError as e:
        print("# numa_balancing unreadable:", e, file=sys.stderr)
    if mode != "none":
        print("# set_mempolicy(%s): node %d rc %d errno %d" % ((mode,) + set_policy(mode)), file=sys.stderr)
    v0 = vmstat()
    sys.argv = sys.argv[2:]
    try:
        runpy.run_path(sys.argv[0], run_name="__main__")
    finally:
        v1 = vmstat()
        print("# vmstat deltas (whole machine):", {k: v1[k] - v0[k] for k in v0}, file=sys.stderr)
