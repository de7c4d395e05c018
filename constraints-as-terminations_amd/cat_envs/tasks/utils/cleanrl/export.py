"""Deployment export of a trained policy: TorchScript ``model.pt`` and ONNX ``model.onnx``.

Reference: scripts/clean_rl/play.py:107-138 traces ``Agent.forward`` (cleanrl/ppo.py:121-123 =
frozen observation normaliser -> ``actor_mean`` MLP, deterministic action) with a (1, D) dummy
input and writes ``exported/model.onnx`` (opset 16, input "input", output "output") and
``exported/model.pt``.

The training-side ``Agent`` of this package runs on HIP kernels over a flat parameter buffer, which a
tracer cannot see, and the robot that consumes the export has no MI355X.  The export therefore
rebuilds the deterministic policy from the checkpoint's ``state_dict`` (the reference's 23-key wire
format) as a plain torch-op module on the CPU and traces *that*; the ONNX file is serialised
directly (``onnx``/``onnxscript`` are not installed in this image, and the graph is six node kinds).
Neither path touches the HIP library or the training hot path: this is file-format host code.
"""
from __future__ import annotations

import os
import struct
from collections import OrderedDict
from typing import Dict, Iterable, List, Mapping, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

OBS_EPS = 1e-8           # RunningMeanStd epsilon, cleanrl/ppo.py:13,37
ONNX_OPSET = 16          # play.py:120
ONNX_IR_VERSION = 8      # IR version that introduced opset 16 (onnx 1.11)


def _cpu_state(sd_or_agent) -> "OrderedDict[str, torch.Tensor]":
    sd = sd_or_agent.state_dict() if hasattr(sd_or_agent, "state_dict") else sd_or_agent
    return OrderedDict((k, torch.as_tensor(v).detach().to("cpu")) for k, v in sd.items())


def actor_layers(sd: Mapping[str, torch.Tensor]) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """[(weight (out,in), bias (out,))...] of ``actor_mean`` in layer order (Linear at even indices)"""
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("actor_mean.") and k.endswith(".weight")})
    if not idx:
        raise KeyError("state_dict has no actor_mean.<i>.weight entries")
    return [(sd[f"actor_mean.{i}.weight"].float(), sd[f"actor_mean.{i}.bias"].float()) for i in idx]


class DeployedPolicy(nn.Module):
    """``Agent.forward(x, deterministic=True)`` as torch ops: ``actor_mean((x - mean) / sqrt(var + eps))``"""

    def __init__(self, state_dict):
        super().__init__()
        sd = _cpu_state(state_dict)
        self.register_buffer("running_mean", sd["obs_rms.running_mean"].float().clone())
        self.register_buffer("running_var", sd["obs_rms.running_var"].float().clone())
        mods: List[nn.Module] = []
        layers = actor_layers(sd)
        for n, (w, b) in enumerate(layers):
            lin = nn.Linear(w.shape[1], w.shape[0])
            with torch.no_grad():
                lin.weight.copy_(w)
                lin.bias.copy_(b)
            mods.append(lin)
            if n + 1 < len(layers):
                mods.append(nn.ELU())
        self.actor_mean = nn.Sequential(*mods)
        self.obs_dim = int(layers[0][0].shape[1])
        self.act_dim = int(layers[-1][0].shape[0])
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = (x - self.running_mean) / torch.sqrt(self.running_var + OBS_EPS)
        return self.actor_mean(x)


def export_policy_as_jit(state_dict, path: str) -> str:
    """TorchScript trace with a (1, D) dummy input, like play.py:133-135"""
    pol = DeployedPolicy(state_dict).eval()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with torch.no_grad():
        traced = torch.jit.trace(pol, torch.zeros(1, pol.obs_dim))
    traced.save(path)
    return path


# --------------------------------------------------------------------------------------------------
# minimal protobuf writer for the ONNX messages used below (field numbers from onnx.proto3)
def _varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _f_varint(field: int, v: int) -> bytes:
    return _varint(field << 3) + _varint(v)


def _f_bytes(field: int, b: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(b)) + b


def _f_str(field: int, s: str) -> bytes:
    return _f_bytes(field, s.encode("utf-8"))


def _f_float(field: int, f: float) -> bytes:
    return _varint((field << 3) | 5) + struct.pack("<f", f)


_FLOAT = 1  # TensorProto.DataType.FLOAT


def _tensor(name: str, arr: np.ndarray) -> bytes:
    arr = np.ascontiguousarray(arr, dtype="<f4")
    m = b"".join(_f_varint(1, int(d)) for d in arr.shape)       # dims
    m += _f_varint(2, _FLOAT)                                     # data_type
    m += _f_str(8, name)                                          # name
    m += _f_bytes(9, arr.tobytes())                               # raw_data (little endian)
    return m


def _attr_f(name: str, v: float) -> bytes:
    return _f_str(1, name) + _f_float(2, v) + _f_varint(20, 1)   # type = FLOAT


def _attr_i(name: str, v: int) -> bytes:
    return _f_str(1, name) + _f_varint(3, v) + _f_varint(20, 2)  # type = INT


def _node(op: str, inputs: Sequence[str], outputs: Sequence[str], name: str, attrs: Iterable[bytes] = ()) -> bytes:
    m = b"".join(_f_str(1, i) for i in inputs) + b"".join(_f_str(2, o) for o in outputs)
    m += _f_str(3, name) + _f_str(4, op)
    m += b"".join(_f_bytes(5, a) for a in attrs)
    return m


def _value_info(name: str, shape: Sequence) -> bytes:
    dims = b""
    for d in shape:
        dim = _f_str(2, d) if isinstance(d, str) else _f_varint(1, int(d))   # dim_param | dim_value
        dims += _f_bytes(1, dim)
    tensor_type = _f_varint(1, _FLOAT) + _f_bytes(2, dims)
    return _f_str(1, name) + _f_bytes(2, _f_bytes(1, tensor_type))


def onnx_graph(state_dict) -> Tuple[List[dict], "OrderedDict[str, np.ndarray]", int, int]:
    """(nodes, initializers, obs_dim, act_dim) of the deterministic policy.  The constant
    ``sqrt(var + eps)`` is folded like ``do_constant_folding=True`` does (play.py:121)."""
    sd = _cpu_state(state_dict)
    init: "OrderedDict[str, np.ndarray]" = OrderedDict()
    init["obs_rms.running_mean"] = sd["obs_rms.running_mean"].float().numpy()
    init["obs_rms.running_std"] = torch.sqrt(sd["obs_rms.running_var"].float() + OBS_EPS).numpy()
    nodes = [dict(op="Sub", inputs=["input", "obs_rms.running_mean"], outputs=["obs_centered"], attrs={}),
             dict(op="Div", inputs=["obs_centered", "obs_rms.running_std"], outputs=["obs_normalized"], attrs={})]
    cur = "obs_normalized"
    layers = actor_layers(sd)
    for n, (w, b) in enumerate(layers):
        wn, bn = f"actor_mean.{2 * n}.weight", f"actor_mean.{2 * n}.bias"
        init[wn], init[bn] = w.numpy(), b.numpy()
        last = n + 1 == len(layers)
        out = "output" if last else f"actor_mean.{2 * n}.out"
        nodes.append(dict(op="Gemm", inputs=[cur, wn, bn], outputs=[out],
                          attrs=dict(alpha=1.0, beta=1.0, transB=1)))
        cur = out
        if not last:
            out = f"actor_mean.{2 * n + 1}.out"
            nodes.append(dict(op="Elu", inputs=[cur], outputs=[out], attrs=dict(alpha=1.0)))
            cur = out
    return nodes, init, int(layers[0][0].shape[1]), int(layers[-1][0].shape[0])


def export_policy_as_onnx(state_dict, path: str, batch=1) -> str:
    """ONNX ModelProto (opset 16) with input "input" (batch, D) and output "output" (batch, A);
    ``batch`` may be a string to declare a symbolic batch axis (the reference exports batch 1)."""
    nodes, init, d, a = onnx_graph(state_dict)
    g = b""
    for n, nd in enumerate(nodes):
        attrs = [(_attr_f if isinstance(v, float) else _attr_i)(k, v) for k, v in nd["attrs"].items()]
        g += _f_bytes(1, _node(nd["op"], nd["inputs"], nd["outputs"], f"/{nd['op']}_{n}", attrs))
    g += _f_str(2, "main_graph")
    for name, arr in init.items():
        g += _f_bytes(5, _tensor(name, arr))
    g += _f_bytes(11, _value_info("input", (batch, d)))
    g += _f_bytes(12, _value_info("output", (batch, a)))
    model = _f_varint(1, ONNX_IR_VERSION) + _f_str(2, "cat_envs") + _f_str(3, "1")
    model += _f_bytes(7, g)
    model += _f_bytes(8, _f_varint(2, ONNX_OPSET))               # opset_import: default domain, version 16
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(model)
    return path


def export_policy(state_dict, directory: str) -> Dict[str, str]:
    """``<directory>/model.onnx`` + ``<directory>/model.pt`` (play.py:111-135)"""
    return {"onnx": export_policy_as_onnx(state_dict, os.path.join(directory, "model.onnx")),
            "jit": export_policy_as_jit(state_dict, os.path.join(directory, "model.pt"))}
