"""Seeded synthetic inputs shared by tests, smoke and bench (SURVEY 8d generator)."""
import numpy as np
import torch


def synth_noblank(seed, T, B, C, S, var_T=False, int64=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, B, C, generator=g)
    L = torch.randint(1, S + 1, (B,), generator=g)
    lab = torch.randint(0, C, (B, S), generator=g, dtype=torch.int32)
    lab[torch.arange(S)[None, :] >= L[:, None]] = -1
    if var_T:
        Tb = torch.maximum(torch.randint(min(S, T), T + 1, (B,), generator=g), L)
    else:
        Tb = torch.full((B,), T, dtype=torch.int64)
    return x, (lab.long() if int64 else lab), Tb.long(), L.long()


def synth_binary(seed, T, B, C, S, var_T=False, density=0.05):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, B, C, generator=g)
    L = torch.randint(1, S + 1, (B,), generator=g)
    y = (torch.rand(B, S, C, generator=g) < density).float()
    y[(torch.arange(S)[None, :] >= L[:, None])] = 0.0
    if var_T:
        Tb = torch.maximum(torch.randint(min(S, T), T + 1, (B,), generator=g), L)
    else:
        Tb = torch.full((B,), T, dtype=torch.int64)
    return x, y, Tb.long(), L.long()


def synth_blank(seed, T, B, C, S, var_T=False):
    g = torch.Generator().manual_seed(seed)
    lp = torch.randn(T, B, C, generator=g).log_softmax(2)
    tgt = torch.randint(1, C, (B, S), generator=g)
    L = torch.randint(1, S + 1, (B,), generator=g)
    if var_T:
        Tb = torch.randint(min(2 * S + 1, T), T + 1, (B,), generator=g)
    else:
        Tb = torch.full((B,), T, dtype=torch.int64)
    return lp, tgt, Tb.long(), L.long()


def np_(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
