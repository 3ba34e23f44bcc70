"""Lu2022 transformer-based coder: `NIC` and its Residual Swin Transformer Blocks, with the module / parameter names of the
reference's models/nic_cvt.py and models/layers.py so its checkpoints load and `quantization` can wrap it.

This is the floating-point model the PTQ flow receives (plain torch modules, like the CompressAI models in this package); the
quantised wrappers that run on the HIP kernels are quantization/quant_block.py: QuantRSTB and friends.
Topology facts that decide the calibration schedule (nic_cvt.py:42-43, 49-219): twelve RSTBs with depths
(2,4,6,2,2,2,2,2,2,6,4,2) and heads (4,8,8,16,16,16,16,16,16,8,8,4); g_a/h_a resample with stride-2 convs (5x5 first, 3x3
after), h_s/g_s with stride-2 transposed convs (5x5 last); the hyper coders use half the window size; children are registered
in the order g_a0..7, h_a0..3, h_s0..3, g_s0..7, entropy_bottleneck, gaussian_conditional, context_prediction,
entropy_parameters."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .entropy import EntropyBottleneck, GaussianConditional
from .layers import MaskedConv2d

DEPTHS = (2, 4, 6, 2, 2, 2, 2, 2, 2, 6, 4, 2)
HEADS = (4, 8, 8, 16, 16, 16, 16, 16, 16, 8, 8, 4)


class DropPath(nn.Module):
    """Stochastic depth; identity in eval mode (the only mode the PTQ flow runs the FP model in)."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        keep = 1.0 - self.p
        m = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * m / keep


class PatchEmbed(nn.Module):
    def forward(self, x):                      # [B, C, H, W] -> [B, H*W, C]
        return x.flatten(2).transpose(1, 2)


class PatchUnEmbed(nn.Module):
    def forward(self, x, x_size):              # [B, H*W, C] -> [B, C, H, W]
        return x.transpose(1, 2).reshape(x.shape[0], -1, x_size[0], x_size[1])


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


def window_partition(x, ws):
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)


def window_reverse(windows, ws, H, W):
    B = windows.shape[0] // ((H // ws) * (W // ws))
    return windows.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def relative_position_index(ws_h, ws_w):
    gh, gw = torch.meshgrid(torch.arange(ws_h), torch.arange(ws_w), indexing="ij")
    pos = torch.stack([gh.reshape(-1), gw.reshape(-1)])
    rel = pos[:, :, None] - pos[:, None, :]
    return (rel[0] + ws_h - 1) * (2 * ws_w - 1) + rel[1] + ws_w - 1


def shift_mask(H, W, ws, shift):
    """0 / -100 mask [nW, N, N] between tokens that the cyclic shift brought together from different image regions."""
    ids = torch.zeros(1, H, W, 1)
    bands = (slice(0, -ws), slice(-ws, -shift), slice(-shift, None))
    for a, hs in enumerate(bands):
        for b, wsl in enumerate(bands):
            ids[:, hs, wsl, :] = 3 * a + b
    mw = window_partition(ids, ws).view(-1, ws * ws)
    diff = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.zeros_like(diff).masked_fill(diff != 0, -100.0)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        wh, ww = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wh - 1) * (2 * ww - 1), num_heads))
        self.register_buffer("relative_position_index", relative_position_index(wh, ww))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.softmax = nn.Softmax(dim=-1)

    def position_bias(self):
        n = self.window_size[0] * self.window_size[1]
        return self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, -1).permute(2, 0, 1).contiguous()

    def forward(self, x, mask=None):
        B_, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B_, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4).unbind(0)
        attn = (q * self.scale) @ k.transpose(-2, -1) + self.position_bias().unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            attn = (attn.view(B_ // nW, nW, self.num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, self.num_heads, N, N)
        attn = self.attn_drop(self.softmax(attn))
        return self.proj_drop(self.proj((attn @ v).transpose(1, 2).reshape(B_, N, C)))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, tuple(input_resolution), num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        if min(self.input_resolution) <= self.window_size:          # one window covers the map: no partition, no shift
            self.shift_size, self.window_size = 0, min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, (self.window_size, self.window_size), num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.register_buffer("attn_mask", self.calculate_mask(self.input_resolution) if self.shift_size > 0 else None)

    def calculate_mask(self, x_size):
        return shift_mask(x_size[0], x_size[1], self.window_size, self.shift_size)

    def forward(self, x, x_size):
        H, W = x_size
        B, L, C = x.shape
        h = self.norm1(x).view(B, H, W, C)
        if self.shift_size > 0:
            h = torch.roll(h, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2))
        win = window_partition(h, self.window_size).view(-1, self.window_size * self.window_size, C)
        mask = self.attn_mask if tuple(x_size) == self.input_resolution else \
            (self.calculate_mask(x_size).to(x.device) if self.shift_size > 0 else None)
        h = window_reverse(self.attn(win, mask=mask).view(-1, self.window_size, self.window_size, C), self.window_size, H, W)
        if self.shift_size > 0:
            h = torch.roll(h, shifts=(self.shift_size, self.shift_size), dims=(1, 2))
        x = x + self.drop_path(h.view(B, H * W, C))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, use_checkpoint=False):
        super().__init__()
        self.dim, self.input_resolution, self.depth, self.use_checkpoint = dim, tuple(input_resolution), depth, use_checkpoint
        self.blocks = nn.ModuleList(
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio,
                                 qkv_bias, qk_scale, drop, attn_drop, drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer) for i in range(depth))

    def forward(self, x, x_size):
        for blk in self.blocks:
            x = blk(x, x_size)
        return x


class RSTB(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, use_checkpoint=False):
        super().__init__()
        self.dim, self.input_resolution = dim, tuple(input_resolution)
        self.residual_group = BasicLayer(dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale,
                                         drop, attn_drop, drop_path, norm_layer, use_checkpoint)
        self.patch_embed = PatchEmbed()
        self.patch_unembed = PatchUnEmbed()

    def forward(self, x, x_size):
        return self.patch_unembed(self.residual_group(self.patch_embed(x), x_size), x_size) + x


# (name, kind, spatial divisor of the stage OUTPUT relative to the image)
_G_A = (("g_a0", "conv5", 2), ("g_a1", "rstb", 2), ("g_a2", "conv3", 4), ("g_a3", "rstb", 4), ("g_a4", "conv3", 8),
        ("g_a5", "rstb", 8), ("g_a6", "conv3", 16), ("g_a7", "rstb", 16))
_H_A = (("h_a0", "conv3", 32), ("h_a1", "rstb", 32), ("h_a2", "conv3", 64), ("h_a3", "rstb", 64))
_H_S = (("h_s0", "rstb", 64), ("h_s1", "tconv3", 32), ("h_s2", "rstb", 32), ("h_s3", "tconv3", 16))
_G_S = (("g_s0", "rstb", 16), ("g_s1", "tconv3", 8), ("g_s2", "rstb", 8), ("g_s3", "tconv3", 4), ("g_s4", "rstb", 4),
        ("g_s5", "tconv3", 2), ("g_s6", "rstb", 2), ("g_s7", "tconv5", 1))


class NIC(nn.Module):
    def __init__(self, config):
        super().__init__()
        c = config
        E, M, ws = c["embed_dim"], c["latent_dim"], c["window_size"]
        H, W = c["height"], c["width"]
        self.M = M
        n_enc = sum(DEPTHS[:6])
        enc_rates = [v.item() for v in torch.linspace(0, c.get("drop_path_rate", 0.0), n_enc)]
        dec_rates = enc_rates[::-1]
        # channel width per stage output; the analysis path widens to M at g_a6, the synthesis path narrows back at g_s1
        width = {"g_a6": M, "g_a7": M, "h_s3": 2 * M, "g_s0": M, "g_s7": c["in_chans"]}
        rstb_no = 0
        prev = c["in_chans"]
        for coder in (_G_A, _H_A, _H_S, _G_S):
            if coder is _H_A or coder is _G_S:
                prev = M
            elif coder is _H_S:
                prev = E
            for name, kind, div in coder:
                out = width.get(name, E)
                if kind == "rstb":
                    i = rstb_no
                    rstb_no += 1
                    lo = sum(DEPTHS[:i]) if i < 6 else sum(DEPTHS[6:i])
                    rates = (enc_rates if i < 6 else dec_rates)[lo:lo + DEPTHS[i]]
                    win = ws // 2 if name[0] == "h" else ws
                    mod = RSTB(dim=prev, input_resolution=(H // div, W // div), depth=DEPTHS[i], num_heads=HEADS[i],
                               window_size=win, mlp_ratio=c.get("mlp_ratio", 2.0), qkv_bias=c.get("qkv_bias", True),
                               qk_scale=c.get("qk_scale"), drop=c.get("drop_rate", 0.0), attn_drop=c.get("attn_drop_rate", 0.0),
                               drop_path=rates, use_checkpoint=c.get("use_checkpoint", False))
                    out = prev
                elif kind.startswith("conv"):
                    k = int(kind[-1])
                    mod = nn.Conv2d(prev, out, kernel_size=k, stride=2, padding=k // 2)
                else:
                    k = int(kind[-1])
                    mod = nn.ConvTranspose2d(prev, out, kernel_size=k, stride=2, padding=k // 2, output_padding=1)
                setattr(self, name, mod)
                prev = out
        self.entropy_bottleneck = EntropyBottleneck(E)
        self.gaussian_conditional = GaussianConditional(None)
        self.context_prediction = MaskedConv2d(M, M * 2, kernel_size=5, padding=2, stride=1)
        self.entropy_parameters = nn.Sequential(nn.Conv2d(M * 12 // 3, M * 10 // 3, 1), nn.LeakyReLU(inplace=True),
                                                nn.Conv2d(M * 10 // 3, M * 8 // 3, 1), nn.LeakyReLU(inplace=True),
                                                nn.Conv2d(M * 8 // 3, M * 6 // 3, 1))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.zeros_(m.bias)
            nn.init.ones_(m.weight)

    def _run(self, stages, x, x_size):
        for name, kind, div in stages:
            mod = getattr(self, name)
            x = mod(x, (x_size[0] // div, x_size[1] // div)) if kind == "rstb" else mod(x)
        return x

    def g_a(self, x, x_size=None):
        return self._run(_G_A, x, x.shape[2:4] if x_size is None else x_size)

    def g_s(self, x, x_size=None):
        return self._run(_G_S, x, (x.shape[2] * 16, x.shape[3] * 16) if x_size is None else x_size)

    def h_a(self, x, x_size=None):
        return self._run(_H_A, x, (x.shape[2] * 16, x.shape[3] * 16) if x_size is None else x_size)

    def h_s(self, x, x_size=None):
        return self._run(_H_S, x, (x.shape[2] * 64, x.shape[3] * 64) if x_size is None else x_size)

    def forward(self, x):
        x_size = (x.shape[2], x.shape[3])
        y = self.g_a(x, x_size)
        z = self.h_a(y, x_size)
        z_hat, z_lik = self.entropy_bottleneck(z)
        hyper = self.h_s(z_hat, x_size)
        y_hat = self.gaussian_conditional.quantize(y, "noise" if self.training else "dequantize")
        ctx = self.context_prediction(y_hat)
        scales, means = self.entropy_parameters(torch.cat((hyper, ctx), dim=1)).chunk(2, 1)
        _, y_lik = self.gaussian_conditional(y, scales, means=means)
        return {"x_hat": self.g_s(y_hat, x_size), "likelihoods": {"y": y_lik, "z": z_lik}}
