# A/B of a render change inside the full bench: the library of the previous commit (stillleben_amd/lib/libslhip_head.so, built
# by hand from `git archive HEAD`) against the working tree's
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - <<P
import json
try:
    d=[json.loads(l) for l in open("gpurun_out/ab_$name.json") if l.startswith("{")][-1]
    print("$name", round(d["value"]), round(d["ms_per_step"]), round(d["roofline_render"]["ms_per_launch"],1), {k:round(x["ms_per_launch"],2) for k,x in d["roofline_render"]["per_kernel"].items()})
except Exception as e:
    print("$name failed", e); print(open("gpurun_out/ab_$name.err").read()[-600:])
P
}
[ -n "$AB_HEAD" ] && [ -f stillleben_amd/lib/libslhip_head.so ] && run head SLHIP_LIB=$PWD/stillleben_amd/lib/libslhip_head.so
run new
