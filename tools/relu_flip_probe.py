"""CPU experiment behind the tolerance of tests/test_hip_train.py::test_train_sequence_encoder_wide_vs_oracle: add
f32-rounding-sized noise to every conv output of the f64 oracle and watch the encoder gradients move by 1e-3..5e-3 in
steps (one ReLU mask flip at a time) - the conditioning of the problem, not of an implementation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import protnote_oracle as O
from tests.helpers import random_encoder_sd, random_head_sd
torch.set_num_threads(8)
C=1100
gen = torch.Generator().manual_seed(41)
ecfg = dict(num_labels=8, input_channels=20, output_channels=C, kernel_size=9, dilation_base=3, num_resnet_blocks=5, bottleneck_factor=0.5)
sd = {"sequence_encoder." + k: v for k, v in random_encoder_sd(ecfg, gen).items()}
sd.update(random_head_sd(gen, C, 1024, 64, 128, 2, 128, 2))
lens = torch.tensor([100,37,64,100])
B, Lmax, NL = len(lens), 100, 24
ids = torch.randint(0, 20, (B, Lmax), generator=gen)
x = torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float().contiguous()
lab = torch.randn(NL, 1024, generator=gen); y = (torch.rand(B, NL, generator=gen) < 0.3).float()
osd = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
run = lambda: O.train_step(osd, x.double(), lens, lab.double(), y.double(), loss="BCE", apply_update=False, train_sequence_encoder=True)
_, l64, g64, _ = run()
orig = O.F.conv1d
for noise in (3e-7, 1e-6, 4e-6):
    ng = torch.Generator().manual_seed(1)
    def noisy(x, w, b, **kw):
        z = orig(x, w, b, **kw)
        # rounding-like perturbation relative to the rms of the output
        return z + noise * z.detach().pow(2).mean().sqrt() * torch.randn(z.shape, generator=ng, dtype=z.dtype)
    O.F.conv1d = noisy
    _, l, g, _ = run()
    O.F.conv1d = orig
    print("noise", noise)
    for n in ["sequence_encoder.conv1.weight"] + [f"sequence_encoder.resnet_blocks.{i}.masked_conv1.weight" for i in range(5)]:
        print(f"  {n:60s} {( (g[n]-g64[n]).norm()/g64[n].norm()).item():.2e}")
