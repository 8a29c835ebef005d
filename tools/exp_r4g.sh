set -x
timeout 900 python -m pytest tests -x -q -m gpu -k "pictures_in_one_call or both_textures or decompress" 2>&1 | tail -15
timeout 600 python bench.py --no-cpu-baseline --c5-frames 0 2>&1 | tail -1 > gpurun_out/b_rgba.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b_rgba.json').read())
print(d.get("value"), d.get("frames_to_rgba"), d.get("texture_to_rgba"), d["kernels"])
PY
