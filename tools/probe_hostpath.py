"""Dev probe: where the host-pointer encode path spends its time."""
import time, os, sys
import torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hap_amd
from hap_amd import synth
ctx = hap_amd.Context(0)
w, h, fmt = 7680, 4320, 1
img = synth.rgba_frame(w, h, 0, device="cuda").cpu().numpy()
tex_bytes = (w // 4) * (h // 4) * 16
dtex = torch.empty(tex_bytes, dtype=torch.uint8, device="cuda")
htex = np.empty(tex_bytes, dtype=np.uint8)
cap = hap_amd.HapMaxEncodedLength([tex_bytes], [fmt], [24])
hframe = np.empty(cap, dtype=np.uint8); hframe.fill(0)
torch.cuda.synchronize()
def t(name, fn, reps=3):
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print("%-44s %.2f ms" % (name, dt * 1e3))
t("compress_rgba host->device", lambda: ctx.compress_rgba(img, w, h, w * 4, fmt, dtex))
t("compress_rgba host->host", lambda: ctx.compress_rgba(img, w, h, w * 4, fmt, htex))
t("encode_frames host tex -> host frame", lambda: ctx.encode_frames([[htex]], [fmt], [1], [24], [hframe], flags=1))
t("encode_frames_rgba host -> host (1 frame)", lambda: ctx.encode_frames_rgba([img], w, h, w * 4, [fmt], [1], [24], [hframe], flags=1))
imgs = [img.copy() for _ in range(4)]; frames = [np.zeros(cap, dtype=np.uint8) for _ in range(4)]
t("encode_frames_rgba host -> host (4 frames)", lambda: ctx.encode_frames_rgba(imgs, w, h, w * 4, [fmt], [1], [24], frames, flags=1))
