"""Generates tests/golden/color_reference_torch.npz from the reference's pure-torch colour conversions
(SealNeRF/color_utils.py: rgb2hsv_torch / hsv2rgb_torch), imported by file path in the build container (torch is its only
import; nothing is written into the reference tree).  Inputs are seeded; the fixture holds inputs and outputs only."""
import importlib.util
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location("ref_color_utils", "/root/reference/SealNeRF/color_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(0)
rgb = rng.random((512, 3), dtype=np.float32)
rgb[:16] = np.round(rgb[:16] * 2) / 2            # ties between channels, greys, pure colours
rgb[16:24] = rgb[16:24, :1]
t = torch.from_numpy(rgb).view(-1, 3, 1)
hsv = ref.rgb2hsv_torch(t.clone())
back = ref.hsv2rgb_torch(hsv.clone())
hsv_in = torch.from_numpy(rng.random((256, 3), dtype=np.float32)).view(-1, 3, 1)
rgb_from = ref.hsv2rgb_torch(hsv_in.clone())
np.savez(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "color_reference_torch.npz"), rgb=rgb, hsv=hsv.view(-1, 3).numpy(), back=back.view(-1, 3).numpy(),
         hsv_in=hsv_in.view(-1, 3).numpy(), rgb_from=rgb_from.view(-1, 3).numpy())
print("ok", float((back.view(-1, 3) - t.view(-1, 3)).abs().max()))
