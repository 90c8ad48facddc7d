import json, sys
for n in sys.argv[1:] or ("c3_fused", "c3_unfused"):
    d = json.load(open("gpurun_out/%s.json" % n))
    print(n, d["value"], d["ms_per_step"], {k: d["config"][k] for k in ("ms_forward_render", "ms_loss_backward", "ms_adam")})
