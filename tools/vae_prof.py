import sys, os, collections, torch
sys.path.insert(0, os.getcwd())
import mobi_amd
from mobi_amd import ops
from bench import build_model
DT = torch.float16 if 'fp16' in sys.argv else torch.bfloat16
mobi_amd.set_engine_dtype(DT)
print(f'storage type {DT}, MOBI_VAE_FP32_TRUNK={os.environ.get("MOBI_VAE_FP32_TRUNK", "(unset: on for fp16)")}')
model = build_model("mobi_nusc_512").cuda()
B = 8
z = torch.randn(B, 4, 64, 64, device="cuda")
x = torch.rand(B, 3, 512, 512, device="cuda") * 2 - 1
xl = torch.rand(B, 2, 512, 512, device="cuda") * 2 - 1
def prof(tag, fn):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        sink = []
        ops.set_profiler(sink)
        fn(); torch.cuda.synchronize()
        ops.set_profiler(None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for kind, fl, a, b, nb, t in sink:
        d = agg.setdefault(kind, [0, 0.0, 0.0]); d[0] += 1; d[1] += a.elapsed_time(b); d[2] += fl
    tot = sum(v[1] for v in agg.values())
    print(f"== {tag}: wall {e0.elapsed_time(e1):.1f} ms, sum of launches {tot:.1f} ms")
    for k, v in agg.items():
        print(f"   {k:18s} n={v[0]:4d} {v[1]:8.2f} ms  {v[2] / max(v[1], 1e-9) / 1e9:8.1f} TFLOP/s")
    rows = sorted(((a.elapsed_time(b), kind, t, fl) for kind, fl, a, b, nb, t in sink), reverse=True)[:12]
    for ms, kind, t, fl in rows:
        print(f"      {ms:7.3f} ms {kind:10s} {fl / max(ms, 1e-9) / 1e9:7.1f} TF  {t}")
prof("camera decode x8", lambda: model.decode_first_stage(z))
prof("lidar decode x8", lambda: model.decode_first_stage(z, module_name="lidar_stage_model"))
prof("camera encode x8", lambda: model.encode_first_stage(x))
prof("lidar encode x8", lambda: model.encode_first_stage(xl, module_name="lidar_stage_model"))
