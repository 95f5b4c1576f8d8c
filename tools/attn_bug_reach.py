"""How much of the synthetic-weight workload did the pre-fix attention re-base bug touch?  Every ring-kernel attention call of one
config-2 B = 2 forward (t = 999 and t = 350) is repeated on the negative-control library (pre-fix code) and compared with the
shipped one."""
import ctypes, json, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import parity_util as PU
from motionclone_amd import spec, lib, ops, build
from motionclone_amd.engine import UNet3DEngine, default_config
lib.load()
dev = torch.device("cuda:0")
cfg = default_config()
sd, _ = spec.synthetic_state_dict(cfg, seed=1234, device=dev)
eng = UNet3DEngine(sd, cfg, dev)
ctl = ctypes.CDLL(build.TRANS_HAZARD_CONTROL_LIB)
ctl.mc_attn_fwd_f16.argtypes = lib.SIGNATURES["mc_attn_fwd_f16"]; ctl.mc_attn_fwd_f16.restype = ctypes.c_int
F, H, W = 16, 64, 64
lat, text, vid, noise = PU.synth_inputs(cfg, F, H, W, dev)
rows = []
for t in (999, 350):
    calls = []
    oa = ops.attn_fwd
    def spy(q, k, v, Nq, Nk, heads, d, nbatch, kv_bdiv=1, scale=None, need_lse=True, out=None):
        o, lse = oa(q, k, v, Nq, Nk, heads, d, nbatch, kv_bdiv=kv_bdiv, scale=scale, need_lse=need_lse, out=out)
        if Nk >= 1024 and d in (40, 80):
            calls.append((q, k, v, Nq, Nk, heads, d, nbatch, kv_bdiv, o))
        return o, lse
    ops.attn_fwd = spy
    eng.forward(lat, t, text, dup=True)
    ops.attn_fwd = oa
    for n, (q, k, v, Nq, Nk, heads, d, nb, kvb, o) in enumerate(calls):
        oc = torch.empty_like(o)
        rc = ctl.mc_attn_fwd_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), v.stride(0), oc.data_ptr(), oc.stride(0),
                                 None, Nq, Nk, heads, d, nb, kvb, float(d ** -0.5), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        diff = (oc != o)
        rowsd = diff.reshape(nb * Nq, heads, d).any(-1)
        rel = float((oc.float() - o.float()).norm() / o.float().norm())
        rows.append(dict(t=t, call=n, Nq=Nq, d=d, frames=nb, elements_differing=int(diff.sum()), of=o.numel(),
                         head_rows_differing=int(rowsd.sum()), of_head_rows=rowsd.numel(), rel_l2=rel))
        print(json.dumps(rows[-1]), flush=True)
tot = sum(r["head_rows_differing"] for r in rows); totn = sum(r["of_head_rows"] for r in rows)
print(json.dumps(dict(summary="pre-fix vs fixed ring attention on the bench's synthetic weights", head_rows_differing=tot, of=totn,
                      fraction=tot / totn, worst_call_rel_l2=max(r["rel_l2"] for r in rows))))
