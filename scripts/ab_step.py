"""Same-box A/B of the training step between library builds: alternating short `bench.py` runs, one process per run (UNET_HIP_LIB selects the
build; `-` = the in-tree library; `name=lib.so;VAR=value;...` also sets UNET_* diagnostics variables for that leg, e.g.
`off=-;UNET_MERGE_BN_FINALIZE=0`).  usage: ab_step.py <bf16|f32> <rounds> name=spec [name=spec ...] [-- extra bench.py flags]
Prints one line per run and the per-build mean / min / max of ms per step."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:]
extra = []
if "--" in argv:
    extra = argv[argv.index("--") + 1:]; argv = argv[:argv.index("--")]
dtype, rounds, builds = argv[0], int(argv[1]), [a.split("=", 1) for a in argv[2:]]
shape = ["--dtype", "bf16", "--channels", "3", "--classes", "4"] if dtype == "bf16" else []
res = {n: [] for n, _ in builds}
for r in range(rounds):
    for name, path in builds:
        env = dict(os.environ)
        env.pop("UNET_HIP_LIB", None)
        path, *sets = path.split(";")
        for kv in sets:
            k, v = kv.split("=", 1); env[k] = v
        if path != "-":
            env["UNET_HIP_LIB"] = os.path.join(ROOT, path)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "8", "--no-extra", "--no-cpu-baseline",
                              "--no-kernel-events"] + shape + extra, env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-400:], flush=True); continue
        d = json.loads(line[-1])
        res[name].append(d["ms_per_step"])
        print("round %d %-10s %8.3f ms  %7.1f img/s  loss %s" % (r, name, d["ms_per_step"], d["value"], d.get("final_loss")), flush=True)
for name, v in res.items():
    if v:
        print("%-10s mean %.3f  min %.3f  max %.3f ms  (n=%d)" % (name, sum(v) / len(v), min(v), max(v), len(v)))
