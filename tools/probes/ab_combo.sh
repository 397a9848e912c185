# A/B of (library, environment, bench arguments) combinations inside the full bench: arguments "<lib name or -> <ENV=..>[,<ENV=..>][|bench args]" ...
for combo in "$@"; do
  bargs=""; case "$combo" in *"|"*) bargs=${combo#*|}; combo=${combo%%|*};; esac
  lib=${combo%% *}; envs=${combo#* }
  args=""; [ "$lib" != "-" ] && args="SLHIP_LIB=$PWD/stillleben_amd/lib/libslhip_$lib.so"
  name=$(echo "$combo$bargs" | tr ' =,/-' '_____')
  env $args $(echo $envs | tr ',' ' ') timeout 300 python bench.py --no-cpu-baseline $bargs > gpurun_out/abc_$name.json 2> gpurun_out/abc_$name.err
  python - <<P
import json
try:
    d=[json.loads(l) for l in open("gpurun_out/abc_$name.json") if l.startswith("{")][-1]
    b=d["breakdown_ms"]
    print("$combo $bargs", round(d["value"]), round(d["ms_per_step"]), "settle", round(b["settle"]), "alone", round(b["settle_alone"]), "render/stream", round(b["render_total_overlapped"]/2), "alone", round(b["render_total_isolated"]))
except Exception as e:
    print("$combo failed", e)
P
done
