"""Soft-attention front end (OTA-style alignment encoder) on the HIP path.

Build-defined spec (the reference snapshot only links the paper,
README.md:50):

    K = Conv1d(C_text -> 2 C_text, k=3) -> ReLU -> Conv1d(2 C_text -> C_att, k=1)      text [B,C_text,T_text]
    Q = Conv1d(C_mel -> 2 C_mel, k=3) -> ReLU -> Conv1d(2 C_mel -> C_mel, k=1) -> ReLU
          -> Conv1d(C_mel -> C_att, k=1)                                                mel  [B,C_mel,T_mel]
    logit[b,i,j] = -temperature * sum_c (Q[b,c,j] - K[b,c,i])^2
    logp = log_softmax over the text axis i (+ log(prior + 1e-8));  layout [B,T_text,T_mel]
    hard = maximum_path(logp, mask)

All arithmetic runs in libaligner_amd.so; torch only owns the buffers.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from . import _lib


def _chk(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must be a GPU tensor")
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_workspaces = _lib.StreamWorkspaces(zero=False)
_conv_workspaces = _lib.StreamWorkspaces(zero=False)      # split activations of the wide conv layers
_prepared: dict = {}          # (weight ptr, shape, device, stream) -> (version, prepared weights, weight)


def _workspace(device, nbytes: int) -> torch.Tensor:
    return _workspaces.get(device, nbytes)             # one per (device, stream)


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _prepared_weights(weight: torch.Tensor, device) -> torch.Tensor:
    """weights -> split bf16 halves in the matrix cores' fragment order (aligner_conv1d_prepare_f32, a few
    microseconds), kept per weight tensor (and stream: no cross-stream ordering) until the tensor is modified in place
    (torch's version counter: the entry is then re-prepared in place, so a training loop does not grow the cache) or
    replaced.  Updates that bypass the counter (`p.data.copy_()`, writes through a storage alias) are NOT seen: call
    invalidate_prepared() after such an update."""
    lib = _lib.load()
    Cout, Cin, K = weight.shape
    key = (weight.data_ptr(), Cout, Cin, K, device, _stream(device))
    ent = _prepared.get(key)
    if ent is None or ent[0] != weight._version:
        nprep = lib.aligner_conv1d_prepared_bytes(Cout, Cin, K)
        if nprep == 0:
            raise ValueError(f"kernel size {K} not supported (1, 3, 5)")
        prep = ent[1] if ent is not None else torch.empty(nprep, dtype=torch.uint8, device=device)
        _lib.check(lib.aligner_conv1d_prepare_f32(weight.data_ptr(), prep.data_ptr(), nprep, Cout, Cin, K, _stream(device)))
        if ent is None and len(_prepared) >= 256:
            _prepared.pop(next(iter(_prepared)))       # oldest entry only
        _prepared[key] = (weight._version, prep, weight)   # holding `weight` keeps its address from being reused
        return prep
    return ent[1]


def conv1d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """y = act(conv1d(x, weight, bias, padding=K//2)); x [B,Cin,T], weight [Cout,Cin,K], K in {1,3,5}."""
    _lib.require_gpu()
    x = _chk(x, "x"); weight = _chk(weight, "weight")
    bias = _chk(bias, "bias") if bias is not None else None
    B, Cin, T = x.shape
    Cout, Cin2, K = weight.shape
    if Cin2 != Cin:
        raise ValueError("channel mismatch")
    y = torch.empty((B, Cout, T), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        prep = _prepared_weights(weight, x.device)
        # wide layers split their activations into a workspace first (csrc/convgemm.hip); narrow ones need none
        nws = lib.aligner_conv1d_workspace_bytes(B, Cin, Cout, T, K)
        ws = _conv_workspaces.get(x.device, nws) if nws else None
        _lib.check(lib.aligner_conv1d_prepared_ws_f32(x.data_ptr(), prep.data_ptr(),
                                                      None if bias is None else bias.data_ptr(), y.data_ptr(),
                                                      None if ws is None else ws.data_ptr(), nws,
                                                      B, Cin, Cout, T, K, int(relu), _stream(x.device)))
    return y


def invalidate_prepared() -> None:
    """Drop every prepared (split-bf16) weight image: call after weight updates that bypass torch's version
    counter (`p.data.copy_()`, EMA through `.data`, writes through a storage alias)."""
    _prepared.clear()


def conv1d_raw(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False
               ) -> torch.Tensor:
    """conv1d() through aligner_conv1d_f32: raw weights, no preparation buffer (slower: every workgroup
    splits its weight tile itself; ALIGNER_CONV_FP32=1 selects the exact-fp32 MFMA kernel)."""
    _lib.require_gpu()
    x = _chk(x, "x"); weight = _chk(weight, "weight")
    bias = _chk(bias, "bias") if bias is not None else None
    B, Cin, T = x.shape
    Cout, Cin2, K = weight.shape
    if Cin2 != Cin:
        raise ValueError("channel mismatch")
    y = torch.empty((B, Cout, T), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().aligner_conv1d_f32(x.data_ptr(), weight.data_ptr(),
                                                  None if bias is None else bias.data_ptr(), y.data_ptr(),
                                                  B, Cin, Cout, T, K, int(relu), _stream(x.device)))
    return y


def soft_attention(keys_enc: torch.Tensor, queries_enc: torch.Tensor, t_x: Optional[torch.Tensor] = None,
                   prior: Optional[torch.Tensor] = None, temperature: float = 0.0005, sim: str = "l2",
                   want_soft: bool = False, out: Optional[torch.Tensor] = None,
                   logp_dtype: torch.dtype = torch.float32, pitched: bool = False
                   ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """logp[b,i,j] (and optionally softmax over text of it) from encoded text/mel.

    pitched: return the log-probs as a [B,T_text,T_mel] VIEW of a buffer whose rows start on whole 128-byte lines
    (pitched_logp(): what align() reads fastest; not contiguous unless T_mel already is a multiple of 128 bytes) where
    the kernel form has a row pitch of its own, contiguous otherwise.

    keys_enc [B,C,T_text], queries_enc [B,C,T_mel] fp32 (channel-major, as the conv
    encoders emit).  Rows i >= t_x[b] are masked to -inf.  logp_dtype torch.bfloat16 writes the log-probs as
    bf16 (half the traffic; align() / maximum_path() read them as they are)."""
    _lib.require_gpu()
    k = _chk(keys_enc, "keys_enc"); q = _chk(queries_enc, "queries_enc")
    B, C, Tx = k.shape
    B2, C2, Ty = q.shape
    if B != B2 or C != C2:
        raise ValueError("keys/queries shape mismatch")
    dev = k.device
    if t_x is not None:
        t_x = t_x.to(device=dev, dtype=torch.int32).contiguous()
    if prior is not None:
        prior = _chk(prior, "prior")
        if tuple(prior.shape) != (B, Tx, Ty):
            raise ValueError("prior must be [B,T_text,T_mel]")
    if out is not None:
        logp_dtype = out.dtype
    if logp_dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("logp_dtype must be torch.float32 or torch.bfloat16")
    if out is None and pitched and prior is None and not want_soft and Tx <= 224 and C in (80, 128) and Ty % 4 == 0:
        try:
            return soft_attention(k, q, t_x=t_x, temperature=temperature, sim=sim, out=pitched_logp(B, Tx, Ty, dev, logp_dtype))
        except _lib.AlignerError as e:          # (a sharp temperature: the exact-product kernel has no row pitch)
            if e.code != _lib.EDOM:
                raise
    logp = out if out is not None else torch.empty((B, Tx, Ty), dtype=logp_dtype, device=dev)
    # `out` may carry a row pitch of its own (a view [:, :, :T_mel] of a [B,T_text,ld] buffer: pitched_logp())
    ld = Ty
    if tuple(logp.shape) == (B, Tx, Ty):
        ld = int(logp.stride(1)) if Tx > 1 else int(logp.stride(0)) if B > 1 else Ty
    if tuple(logp.shape) != (B, Tx, Ty) or logp.stride(2) != 1 or ld < Ty or (B > 1 and logp.stride(0) != Tx * ld):
        raise ValueError("out must be a [B,T_text,T_mel] tensor, contiguous or with a row pitch (pitched_logp())")
    soft = torch.empty((B, Tx, Ty), dtype=torch.float32, device=dev) if want_soft else None
    simc = {"l2": _lib.SIM_L2, "dot": _lib.SIM_DOT}[sim]
    lib = _lib.load()
    with torch.cuda.device(dev):
        ws = _workspace(dev, lib.aligner_softattn_workspace_bytes(B, C, Tx))
        _lib.check(lib.aligner_softattn_ld(
            k.data_ptr(), q.data_ptr(), None if t_x is None else t_x.data_ptr(),
            None if prior is None else prior.data_ptr(), logp.data_ptr(),
            _lib.DT_BF16 if logp_dtype == torch.bfloat16 else _lib.DT_F32, ld,
            None if soft is None else soft.data_ptr(), ws.data_ptr(), ws.numel(),
            B, C, Tx, Ty, float(temperature), simc, _stream(dev)))
    return logp, soft


def pitched_logp(B: int, T_text: int, T_mel: int, device, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """A [B,T_text,T_mel] tensor whose rows start on 128-byte lines (row pitch = T_mel rounded up to 128 bytes): the
    pipeline's own intermediate between soft_attention(out=...) and align().  With T_mel = 1000 fp32 a row is 4000 bytes and
    three quarters of the 128-byte runs the similarity kernel stores (and the search's loaders fetch) straddle two lines; at a
    pitch of 1024 elements none does.  The extra columns are never touched."""
    per = 128 // torch.empty((), dtype=dtype).element_size()
    ld = (T_mel + per - 1) // per * per
    return torch.empty((B, T_text, ld), dtype=dtype, device=device)[:, :, :T_mel]


@dataclass
class AlignmentEncoderParams:
    """Weights of the two conv stacks: lists of (weight [Cout,Cin,K], bias [Cout])."""
    key_proj: List[Tuple[torch.Tensor, torch.Tensor]]
    query_proj: List[Tuple[torch.Tensor, torch.Tensor]]
    temperature: float = 0.0005

    @staticmethod
    def random(c_text: int, c_mel: int, c_att: int, device, seed: int = 0) -> "AlignmentEncoderParams":
        g = torch.Generator(device="cpu").manual_seed(seed)

        def layer(co, ci, k):
            bound = (1.0 / (ci * k)) ** 0.5
            w = (torch.rand((co, ci, k), generator=g) * 2 - 1) * bound
            b = (torch.rand((co,), generator=g) * 2 - 1) * bound
            return w.to(device), b.to(device)

        return AlignmentEncoderParams(
            key_proj=[layer(2 * c_text, c_text, 3), layer(c_att, 2 * c_text, 1)],
            query_proj=[layer(2 * c_mel, c_mel, 3), layer(c_mel, 2 * c_mel, 1), layer(c_att, c_mel, 1)],
        )


def encode(x: torch.Tensor, stack: List[Tuple[torch.Tensor, torch.Tensor]]) -> torch.Tensor:
    """A conv stack (ReLU between the layers) -- in ONE call of the C ABI when every layer has a GEMM form
    (aligner_conv_stack_f32: the input is split once, every k = 1 layer reads the split image its producer wrote; no fp32
    round trip between layers), layer by layer otherwise."""
    _lib.require_gpu()
    lib = _lib.load()
    x = _chk(x, "x")
    B, Cin, T = x.shape
    dev = x.device
    layers = (_lib.ConvLayer * len(stack))()
    keep = []
    c = Cin
    with torch.cuda.device(dev):
        for n, (w, b) in enumerate(stack):
            w = _chk(w, "weight")
            b = _chk(b, "bias") if b is not None else None
            if w.shape[1] != c:
                raise ValueError("channel mismatch")
            prep = _prepared_weights(w, dev)
            keep += [w, b, prep]
            layers[n] = _lib.ConvLayer(prep.data_ptr(), None if b is None else b.data_ptr(), int(w.shape[1]), int(w.shape[0]),
                                       int(w.shape[2]), int(n + 1 < len(stack)))
            c = int(w.shape[0])
        nws = lib.aligner_conv_stack_workspace_bytes(layers, len(stack), B, T) if B > 0 and len(stack) > 0 else 0
        if nws:
            y = torch.empty((B, c, T), dtype=torch.float32, device=dev)
            ws = _conv_workspaces.get(dev, nws)
            _lib.check(lib.aligner_conv_stack_f32(x.data_ptr(), layers, len(stack), y.data_ptr(), ws.data_ptr(), nws, B, T,
                                                  _stream(dev)))
            return y
    for n, (w, b) in enumerate(stack):
        x = conv1d(x, w, b, relu=(n + 1 < len(stack)))
    return x


def alignment_encoder(text_emb: torch.Tensor, mel: torch.Tensor, params: AlignmentEncoderParams,
                      t_x: Optional[torch.Tensor] = None, prior: Optional[torch.Tensor] = None,
                      want_soft: bool = False, pitched: bool = False):
    """text_emb [B,C_text,T_text], mel [B,C_mel,T_mel] -> (logp [B,T_text,T_mel], soft or None).  pitched: see soft_attention()."""
    k = encode(text_emb, params.key_proj)
    q = encode(mel, params.query_proj)
    return soft_attention(k, q, t_x=t_x, prior=prior, temperature=params.temperature, want_soft=want_soft, pitched=pitched)
