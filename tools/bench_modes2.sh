set -o pipefail
for cfg in "--from-base ingest" "--model ViT-B-16 --masks union --masks-packed" "--model ViT-B-16 --masks majority" "--model ViT-B-16 --masks union --masks-compact" "--model ViT-B-16 --masks union --from-base fused"; do
  tag=$(echo "$cfg" | tr -d ' -')
  timeout -k 10 300 python bench.py --no-cpu --steps 5 $cfg > gpurun_out/bm2_$tag.log 2>&1 || { tail -n 8 gpurun_out/bm2_$tag.log; exit 1; }
  python - "gpurun_out/bm2_$tag.log" "$cfg" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(f"[{sys.argv[2]}]", d["ms_per_step"], round(d["value"]/1e3), "k path", d["path_roofline_frac"], "untuned", d.get("untuned", {}).get("ms_per_step"), d["config"]["mask_density"])
PY
done
