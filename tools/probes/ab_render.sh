# A/B of render variants inside the full bench.  AB_LIBS: names of stillleben_amd/lib/libslhip_<name>.so to compare;
# AB_ENVS: ';'-separated environment settings to compare on the working tree's library
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
for n in $AB_LIBS; do run $n SLHIP_LIB=$PWD/stillleben_amd/lib/libslhip_$n.so; done
i=0
IFS=';' read -ra ENVS <<< "$AB_ENVS"
for e in "${ENVS[@]}"; do i=$((i+1)); echo "env$i: $e"; run env$i $e; done
