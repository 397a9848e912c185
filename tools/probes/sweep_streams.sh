run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err; python - <<P
import json
try:
    d=[json.loads(l) for l in open("gpurun_out/sw_$name.json") if l.startswith("{")][-1]
    print("$name", round(d["value"]), round(d["ms_per_step"]), {k:round(x["ms_per_launch"],2) for k,x in d["roofline_render"]["per_kernel"].items()})
except Exception as e:
    print("$name failed", e); print(open("gpurun_out/sw_$name.err").read()[-400:])
P
}
run base
SLHIP_SHADE_TILED=2 run t16x4
run r3 --render-streams 3
run s2r2 --settle-streams 2 --render-streams 2
run c2048 --render-chunk 2048
run c512r3 --render-chunk 512 --render-streams 3
