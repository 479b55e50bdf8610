import json, os, resource, subprocess, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
t0 = time.time()
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "config5", "--dtype", "f32", "--steps", "100",
                      "--warmup", "5", "--no-cpu-baseline", "--no-hbm-resident"], capture_output=True, text=True)
wall = time.time() - t0
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
print(f"bench.py --workload config5 --dtype f32 (one rank's 12.5M-member shard): wall {wall:.1f} s, set-up {d['config']['setup_s_rank0']:.2f} s, "
      f"host max RSS {ru.ru_maxrss / 1e6:.2f} GB, value {d['value']:.3e}, summary {d['summary']['gather_ms']:.2f} ms")
