"""3DGS PLY files for the [N,14] Gaussian layout of the reference (xyz 3 | opacity 1 | scale 3 | rotation 4 | rgb 3).

Mirrors the behaviour of GaussianRenderer.save_ply / load_ply / load_gaussians_from_ply (core/gaussians/gs.py:120-252) without
the `plyfile` / `kiui` dependencies (absent from the ROCm image): a small PLY reader/writer for the `vertex` element
(binary_little_endian or ascii, scalar properties only) lives here.  File I/O is off the rasterizer hot path; this closes
SURVEY 8f rank 4 ("plus 3DGS PLY I/O").

save_ply    prunes Gaussians with opacity < 0.005, and with compatible=True stores the pre-activation values of the original 3DGS
            format: logit(opacity), log(scale + 1e-8), (rgb - 0.5) / C0  (gs.py:133-146); property order x y z f_dc_* opacity scale_* rot_*.
load_ply    the inverse: sigmoid / exp / C0 * f_dc + 0.5 (gs.py:196-199) -> [N,14] on the CPU.
load_gaussians_from_ply   reads a full 3DGS training checkpoint (f_rest_* ignored except for shape), returns activated values with
            normalised quaternions and the colour channels in the reference's B,G,R -> R,G,B order flip (gs.py:243).
"""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

import numpy as np
import torch

C0 = 0.28209479177387814                      # SH band-0 constant (gs.py:254)
_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def write_vertex_ply(path: str, columns: List[Tuple[str, np.ndarray]]) -> None:
    """Binary little-endian PLY with one `vertex` element of float32 properties, in the given column order."""
    n = columns[0][1].shape[0]
    rec = np.empty(n, dtype=[(name, "<f4") for name, _ in columns])
    for name, col in columns:
        rec[name] = np.asarray(col, dtype=np.float32).reshape(n)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"] + [f"property float {name}" for name, _ in columns] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rec.tobytes())


def read_vertex_ply(path: str) -> Dict[str, np.ndarray]:
    """-> {property name: array [N]} of the first element (must be `vertex`, scalar properties)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if count is None and tok[1] == "vertex":
                    count, in_vertex = int(tok[2]), True
                elif count is None:
                    raise ValueError(f"{path}: first element is '{tok[1]}', expected 'vertex'")
                else:
                    in_vertex = False
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported on the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError(f"{path}: PLY header without format / vertex element")
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
            return {name: rows[:, i].astype(np.dtype(t)) for i, (name, t) in enumerate(props)}
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(name, order + t) for name, t in props])
        data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        return {name: np.ascontiguousarray(data[name]) for name, _ in props}


def _numbered(cols: Dict[str, np.ndarray], prefix: str) -> List[str]:
    names = [k for k in cols if re.fullmatch(re.escape(prefix) + r"\d+", k)]
    return sorted(names, key=lambda k: int(k[len(prefix):]))


def save_ply(gaussians: torch.Tensor, path: str, compatible: bool = True) -> None:
    """gaussians [1,N,14] (xyz | opacity | scale | rotation | rgb), gs.py:120-168."""
    assert gaussians.shape[0] == 1, "only support batch size 1"
    g = gaussians[0].detach().float().cpu()
    g = g[g[:, 3] >= 0.005]                                        # prune by opacity (gs.py:132)
    xyz, opacity, scales, rot, rgb = g[:, 0:3], g[:, 3:4], g[:, 4:7], g[:, 7:11], g[:, 11:14]
    if compatible:                                                 # back to the original format's pre-activation values
        opacity = torch.log(opacity / (1.0 - opacity))             # == kiui.op.inverse_sigmoid
        scales = torch.log(scales + 1e-8)
        rgb = (rgb - 0.5) / C0
    cols = [("x", xyz[:, 0]), ("y", xyz[:, 1]), ("z", xyz[:, 2])]
    cols += [(f"f_dc_{i}", rgb[:, i]) for i in range(rgb.shape[1])]
    cols += [("opacity", opacity[:, 0])]
    cols += [(f"scale_{i}", scales[:, i]) for i in range(3)] + [(f"rot_{i}", rot[:, i]) for i in range(4)]
    write_vertex_ply(path, [(n, c.numpy()) for n, c in cols])


def load_ply(path: str, compatible: bool = True) -> torch.Tensor:
    """-> [N,14] float32 CPU tensor (gs.py:170-201)."""
    c = read_vertex_ply(path)
    n = c["x"].shape[0]
    print("Number of points at loading : ", n)
    scale_names = [k for k in c if k.startswith("scale_")]         # file order, like the reference
    rot_names = [k for k in c if k.startswith("rot_")]
    parts = [np.stack([c["x"], c["y"], c["z"]], 1), c["opacity"][:, None], np.stack([c[k] for k in scale_names], 1),
             np.stack([c[k] for k in rot_names], 1), np.stack([c["f_dc_0"], c["f_dc_1"], c["f_dc_2"]], 1)]
    g = torch.from_numpy(np.concatenate([p.astype(np.float64) for p in parts], 1)).float()
    if compatible:
        g[:, 3:4] = torch.sigmoid(g[:, 3:4])
        g[:, 4:7] = torch.exp(g[:, 4:7])
        g[:, 11:] = C0 * g[:, 11:] + 0.5
    return g


def load_gaussians_from_ply(path: str) -> torch.Tensor:
    """A full 3DGS checkpoint -> [N,14] activated Gaussians (gs.py:203-252): sigmoid opacity, exp scale, normalised rotation,
    SH band 0 -> rgb with the channel order reversed exactly like the reference does."""
    c = read_vertex_ply(path)
    f32 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), dtype=torch.float)
    xyz = f32(np.stack([c["x"], c["y"], c["z"]], 1))
    dc = np.stack([c["f_dc_0"], c["f_dc_1"], c["f_dc_2"]], 1).astype(np.float64)
    rest = _numbered(c, "f_rest_")
    if rest and len(rest) != 3 * ((3 + 1) ** 2 - 1):
        raise ValueError(f"{path}: expected {3 * 15} f_rest_* properties of a degree-3 checkpoint, found {len(rest)}")
    scales = torch.exp(f32(np.stack([c[k] for k in _numbered(c, "scale_")], 1)))
    rots = torch.nn.functional.normalize(f32(np.stack([c[k] for k in _numbered(c, "rot_")], 1)))
    colors = f32((dc * C0 + 0.5)[:, [2, 1, 0]])
    opacity = torch.sigmoid(f32(c["opacity"][:, None]))
    return torch.cat([xyz, opacity, scales, rots, colors], dim=1)
