"""Debug helper: obs / prompt tokens of the e384_long case for a given (vit_prune_last, vit_chunk), saved for a bitwise comparison
between processes started with different VIMA_VIT_ATTN_LDS:  python scripts/vit_lds_check.py TAG prune chunk"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle.cases import build_case  # noqa: E402
from vima_testing import synthetic as syn  # noqa: E402
from tests.gpu_common import loaded_policy  # noqa: E402

tag, prune, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg, wseed, prompts, obs, actions = build_case("e384_long")
sd = syn.make_state_dict(cfg, wseed)
pol = loaded_policy(cfg, sd, "bf16", dual_stream=0, vit_prune_last=prune, vit_chunk=chunk)
p = syn.to_device(prompts, "cuda:0")
o = syn.to_device(obs, "cuda:0")
ptok, _ = pol.forward_prompt_assembly(p)
otok, _ = pol.forward_obs_token(o)
torch.save({"ptok": ptok.cpu(), "otok": otok.cpu()}, f"gpurun_out/vitchk_{tag}.pt")
print(tag, ptok.shape, otok.shape, float(ptok.abs().sum()), float(otok.abs().sum()))
