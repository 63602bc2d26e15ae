"""Compare a tensor with a fixture entry written by tools/gen_golden.py (`put`)."""
import numpy as np
import torch

from oracle.closed_form import checksum


def check_entry(fx, key, t, rtol, atol, what=""):
    t = t.detach().float().cpu()
    if key + "/full" in fx:
        ref = torch.from_numpy(fx[key + "/full"])
        assert tuple(ref.shape) == tuple(t.shape), f"{what}{key}: shape {tuple(t.shape)} vs {tuple(ref.shape)}"
        torch.testing.assert_close(t, ref, rtol=rtol, atol=atol, msg=lambda m: f"{what}{key}: {m}")
        return
    s, v = checksum(t)
    ref_s, ref_v = fx[key + "/sum"], fx[key + "/samples"]
    np.testing.assert_allclose(v, ref_v, rtol=rtol, atol=atol, err_msg=f"{what}{key} samples")
    # abs-sum is a well-conditioned whole-tensor check; the plain sum is checked relative to it
    n = max(t.numel(), 1)
    assert abs(s[1] - ref_s[1]) <= rtol * abs(ref_s[1]) + atol * n, f"{what}{key} abs-sum {s[1]} vs {ref_s[1]}"
    assert abs(s[0] - ref_s[0]) <= rtol * abs(ref_s[1]) + atol * n, f"{what}{key} sum {s[0]} vs {ref_s[0]}"


def entries(fx, prefix):
    """Names stored under `prefix/` (without the /full,/sum,/samples suffix)."""
    out = set()
    for k in fx.files:
        if k.startswith(prefix + "/"):
            out.add(k[len(prefix) + 1:].rsplit("/", 1)[0])
    return sorted(out)
