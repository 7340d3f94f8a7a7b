"""The ViT oracle (oracle/vit_oracle.py) against golden vectors captured from the unmodified reference.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import vit_oracle as vorc
from tests.helpers import golden_files, load_case


@pytest.mark.parametrize("path", golden_files("f7_") + golden_files("f8_"), ids=lambda p: p.split("/")[-1][:-4])
def test_vit_oracle_matches_reference(path):
    z, sd = load_case(path)
    patch, dim, depth, heads, ffn = [int(v) for v in z["cfg"]]
    kind, scale = str(z["kind"]), float(z["scalar"])
    imgs = torch.from_numpy(z["imgs_u8"]).float() / 255.0
    feats = vorc.vit_forward(imgs, sd, patch, depth, heads, scale, kind)
    np.testing.assert_allclose(feats.numpy(), z["feats"], rtol=0, atol=2e-5)
    if kind != "mae_adapter":
        tok = vorc.prepare_tokens(imgs, sd, patch)
        np.testing.assert_allclose(tok[:, ::13, :].numpy(), z["tokens0"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(vorc.block(tok, sd, "blocks.0.", heads, scale)[:, ::13, :].numpy(), z["block0"],
                                   rtol=0, atol=2e-5)
        x = tok
        for i in range(depth - 1):
            x = vorc.block(x, sd, f"blocks.{i}.", heads, scale)
        attn = vorc.block(x, sd, f"blocks.{depth - 1}.", heads, scale, return_attention=True)
        np.testing.assert_allclose(attn[:, :, ::29, :].numpy(), z["last_attn"], rtol=0, atol=1e-6)
