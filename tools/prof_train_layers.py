"""Per-call TFLOP/s of the training step's GEMM launches (forward, data / weight gradients): each C-ABI call timed alone with
HIP events, grouped by shape.  python tools/prof_train_layers.py [B]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpgesture_amd import synth, _lib
from qpgesture_amd.vqvae import VQVAE
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = VQVAE(dict(vel=1, acc=1), 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7)).train()
x = torch.randn((B, 240, 135), device=dev)
for _ in range(2):
    m(x); m.backward()
torch.cuda.synchronize()
rec = collections.OrderedDict()
orig = _lib.call
def timed(name, device, *a):
    if name not in ("qpg_conv1d_bwd_weight_f32", "qpg_conv1d_bwd_data_f32", "qpg_resblock_f32", "qpg_convt_f32", "qpg_conv1d_f32"):
        return orig(name, device, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    r = orig(name, device, *a)
    e1.record(); torch.cuda.synchronize()
    if name == "qpg_resblock_f32":
        Bq, T, dil = a[1], a[2], a[3]
        key = ("resblock", 4, 512, 512, dil, T)
        fl = 2.0 * Bq * T * 4 * 512 * 512
    elif name == "qpg_convt_f32":
        Bq, cx, taps, cout, dil, T_out = a[1], a[3], a[6], a[8], a[12], a[13]
        key = ("convt", taps, cx, cout, dil, T_out)
        fl = 2.0 * Bq * T_out * taps * cx * cout
    elif name == "qpg_conv1d_f32":
        Bq, cin, taps, cout, dil, T_out = a[1], a[3], a[6], a[8], a[12], a[13]
        key = ("conv1d", taps, cin, cout, dil, T_out)
        fl = 2.0 * Bq * T_out * taps * cin * cout
    elif name.endswith("weight_f32"):
        Bq, T_in, cin, taps, cout, dil, T_out = a[1], a[2], a[3], a[5], a[7], a[11], a[12]
        key = ("wgrad", taps, cin, cout, dil, T_out)
        fl = 2.0 * Bq * T_out * taps * cin * cout
    else:
        Bq, T_in, cdy, taps, fcin, dil, T_out = a[1], a[2], a[3], a[5], a[6], a[13], a[14]
        key = ("dgrad", taps, cdy, fcin, dil, T_out)
        fl = 2.0 * Bq * T_out * taps * cdy * fcin
    d = rec.setdefault(key, [0, 0.0, fl])
    d[0] += 1; d[1] += e0.elapsed_time(e1)
    return r
_lib.call = timed
N = 3
for _ in range(N):
    m(x); m.backward()
tot = 0.0
print("| kind | taps | C_a | C_b | dil | T_out | calls/step | ms/call | TFLOP/s | ms/step |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k, (n, ms, fl) in rec.items():
    per = ms / n
    tot += ms / N
    print("| %s | %d | %d | %d | %d | %d | %d | %.3f | %.1f | %.2f |" % (k + (n // N, per, fl / per / 1e9, ms / N)))
print("total %.2f ms per step" % tot)
