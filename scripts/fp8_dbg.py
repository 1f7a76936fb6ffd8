import torch, sys
sys.path.insert(0, '.')
from vima_testing import synthetic as syn
from tests.gpu_common import loaded_policy, max_abs
from tests.test_fp8_gpu import _case
cfg, sd, prompts, obs = _case()
DEV="cuda:0"
p = syn.to_device(prompts, DEV)
pw = loaded_policy(cfg, sd, "fp8w", dual_stream=0)
ref, _ = pw.forward_prompt_assembly(p)
pb = loaded_policy(cfg, sd, "bf16", dual_stream=0)
refb, _ = pb.forward_prompt_assembly(p)
pol = loaded_policy(cfg, sd, "fp8", dual_stream=0)
cal, _ = pol.forward_prompt_assembly(p)
print("cal vs fp8w", max_abs(cal, ref), "scale", ref.abs().max().item(), "fp8w vs bf16", max_abs(ref, refb))
sc = pol.fp8_act_scales()
print(sc)
out, _ = pol.forward_prompt_assembly(p)
print("fp8 vs fp8w", max_abs(out, ref), "mean abs diff", (out-ref).abs().mean().item(), "mean abs ref", ref.abs().mean().item())
d = (out-ref).abs()
print("per-sample max", d.amax(dim=(0,2))[:8])
print("per-position max", d.amax(dim=(1,2))[:16])
