import os, sys
for k in sorted(os.environ):
    if k not in ("_",):
        print(k, "=", os.environ[k])
print("affinity", len(os.sched_getaffinity(0)))
print("nice", os.nice(0))
try:
    import resource
    print("rlimit_memlock", resource.getrlimit(resource.RLIMIT_MEMLOCK))
except Exception as e:
    print(e)
print("cgroup", open("/proc/self/cgroup").read().strip())
print("threads", [l for l in open("/proc/self/status") if l.startswith(("Threads", "Cpus_allowed_list", "voluntary"))])
