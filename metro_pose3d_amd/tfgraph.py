"""TF-free reader (and minimal writer) of frozen TensorFlow GraphDef files (`.pb`).

The reference's `model_path` is a frozen GraphDef written by src/main.py:143-161
(convert_variables_to_constants + TransformGraph) and consumed by inference.py:31-38.  TensorFlow is
neither needed nor available here: a GraphDef is plain protobuf, and only a handful of message
types matter, so this module decodes the wire format directly (public schema of
tensorflow/core/framework/{graph,node_def,attr_value,tensor,tensor_shape,types}.proto).

What is extracted:
  * every `Const` node's tensor (DT_FLOAT / DT_HALF / DT_INT32 / DT_INT64 / DT_STRING), by node name;
  * the slim variables of the backbone + head, found by GRAPH STRUCTURE first: every Conv2D / BiasAdd /
    FusedBatchNorm[V2/V3] node's parameter inputs are followed back through Identity (`.../read`) and Cast nodes to
    the Const that feeds them -- the original fp32 variable, or the fp16 Const that `fold_constants` (main.py:150-157)
    leaves in place of `Cast(variable)` (tfu.py:426-440 casts every trainable variable at use, so in the default
    fp16 export the surviving constant is named after the CAST node, e.g. `.../conv1/Cast/_12__cf__12`, not after
    the variable).  The slim variable name is taken from the chain when one of its nodes still carries it, else from
    the consumer: `<scope of the op>/{weights | biases | gamma, beta, moving_mean, moving_variance}` by input
    position (scopes: reference volumetric.py:158, architectures.py:24, resnet_v2.py:117-136,203-236);
    graphs that are plain bags of constants (the in-repo writer) are read by variable name as before;
  * `centered_stride` (options.py:118) is not stored in a graph; it is inferred from the one place it changes the
    ops: the strided 3x3 conv2 of a bottleneck is `padding='SAME'` when centered and an explicit Pad + 'VALID'
    otherwise (resnet_utils.py:120-135).  Graphs without a strided unit (stride 4) keep the caller's value;
  * `joint_names` / `joint_edges` (main.py:140-141) and, when present, the 5-element Reshape shape
    `[-1, depth, J, S, S]` of volumetric.py:231 from which depth, J_head and the stride follow.

UNTESTED AGAINST A REAL EXPORT: there is no released `.pb` (and no TensorFlow) offline.  The reader is exercised
against graphs encoded with the official protobuf runtime that are shaped like a TF 1.13 export of this model
(variable -> read -> Cast -> [folded] -> Conv2D / BiasAdd / FusedBatchNorm, fp32 and fp16-folded; tests/test_tfgraph.py)
and against the in-repo writer; anything it cannot identify is reported as an error, never guessed around.
"""
from __future__ import annotations

import re
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

# ---- protobuf wire format -------------------------------------------------------------------------
_VARINT, _I64, _LEN, _I32 = 0, 1, 2, 5


def _read_varint(buf: memoryview, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _fields(buf: memoryview):
    """Yields (field_number, wire_type, value) for one message; LEN values are memoryviews."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == _VARINT:
            v, pos = _read_varint(buf, pos)
        elif wt == _I64:
            v = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == _LEN:
            ln, pos = _read_varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
            if len(v) != ln:
                raise ValueError('truncated length-delimited field')
        elif wt == _I32:
            v = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError(f'unsupported wire type {wt}')
        yield fno, wt, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


# tensorflow DataType enum -> numpy
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_STRING, DT_INT64, DT_BOOL, DT_HALF = 1, 2, 3, 7, 9, 10, 19
_NP = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_INT64: np.int64,
       DT_BOOL: np.bool_, DT_HALF: np.float16}


def _parse_shape(buf: memoryview) -> List[int]:
    dims = []
    for fno, wt, v in _fields(buf):
        if fno == 2 and wt == _LEN:                         # TensorShapeProto.dim
            size = 0
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == _VARINT:
                    size = _signed64(v2)
            dims.append(size)
    return dims


def _packed(v, wt, fmt: str, size: int) -> List:
    if wt == _LEN:
        b = bytes(v)
        return list(struct.unpack(f'<{len(b) // size}{fmt}', b))
    return [struct.unpack('<' + fmt, v)[0]]


def _parse_tensor(buf: memoryview) -> np.ndarray:
    dtype, shape, content = 0, [], None
    floats: List[float] = []
    ints: List[int] = []
    halfs: List[int] = []
    strings: List[bytes] = []
    for fno, wt, v in _fields(buf):
        if fno == 1 and wt == _VARINT:
            dtype = v
        elif fno == 2 and wt == _LEN:
            shape = _parse_shape(v)
        elif fno == 4 and wt == _LEN:
            content = bytes(v)
        elif fno == 5:
            floats += _packed(v, wt, 'f', 4)
        elif fno == 6:
            floats += _packed(v, wt, 'd', 8)
        elif fno in (7, 10, 11, 13):                        # int_val, int64_val, bool_val, half_val
            vals = []
            if wt == _LEN:
                mv, p = v, 0
                while p < len(mv):
                    x, p = _read_varint(mv, p)
                    vals.append(_signed64(x))
            else:
                vals.append(_signed64(v))
            (halfs if fno == 13 else ints).extend(vals)
        elif fno == 8 and wt == _LEN:
            strings.append(bytes(v))
    n = int(np.prod(shape)) if shape else 1
    if dtype == DT_STRING:
        arr = np.empty(len(strings), dtype=object)
        arr[:] = strings
        return arr.reshape(shape) if shape and int(np.prod(shape)) == len(strings) else arr
    if dtype not in _NP:
        raise ValueError(f'unsupported tensor dtype {dtype}')
    npdt = _NP[dtype]
    if content is not None and len(content):
        arr = np.frombuffer(content, dtype=npdt).copy()
    elif dtype == DT_HALF:
        arr = np.array(halfs, dtype=np.uint16).view(np.float16)
    elif dtype in (DT_FLOAT, DT_DOUBLE):
        arr = np.array(floats, dtype=npdt)
    else:
        arr = np.array(ints, dtype=npdt)
    if arr.size == 1 and n > 1:
        arr = np.full(n, arr[0], dtype=npdt)                # splat encoding of repeated values
    if arr.size != n:
        raise ValueError(f'tensor has {arr.size} values for shape {shape}')
    return arr.reshape(shape)


class Node:
    __slots__ = ('name', 'op', 'inputs', 'tensor', 'attrs')

    def __init__(self):
        self.name, self.op, self.inputs, self.tensor, self.attrs = '', '', [], None, {}


def _parse_node(buf: memoryview) -> Node:
    node = Node()
    for fno, wt, v in _fields(buf):
        if fno == 1 and wt == _LEN:
            node.name = bytes(v).decode()
        elif fno == 2 and wt == _LEN:
            node.op = bytes(v).decode()
        elif fno == 3 and wt == _LEN:
            node.inputs.append(bytes(v).decode())
        elif fno == 5 and wt == _LEN:                       # map<string, AttrValue> entry
            key, val = None, None
            for f2, w2, v2 in _fields(v):
                if f2 == 1:
                    key = bytes(v2).decode()
                elif f2 == 2:
                    val = v2
            if key == 'value' and val is not None:
                for f3, w3, v3 in _fields(val):
                    if f3 == 8 and w3 == _LEN:              # AttrValue.tensor
                        node.tensor = _parse_tensor(v3)
            elif key is not None and val is not None:
                node.attrs[key] = bytes(val)
    return node


def read_graph(path: str) -> Dict[str, Node]:
    with open(path, 'rb') as f:
        data = memoryview(f.read())
    nodes: Dict[str, Node] = {}
    for fno, wt, v in _fields(data):
        if fno == 1 and wt == _LEN:                         # GraphDef.node
            n = _parse_node(v)
            nodes[n.name] = n
    if not nodes:
        raise ValueError(f'{path}: no nodes found (not a GraphDef?)')
    return nodes


# ---- GraphDef -> (ModelSpec, slim variable dict) ------------------------------------------------------
_VAR_RE = re.compile(r'^(MainPart/resnet_v2_(50|101)/.*?/(weights|biases|gamma|beta|moving_mean|moving_variance))(/.*)?$')
_KINDS = ('weights', 'biases', 'gamma', 'beta', 'moving_mean', 'moving_variance')
_FOLD_RE = re.compile(r'/_\d+__cf__\d+$')                      # suffix fold_constants gives the constants it creates
_PASS_RE = re.compile(r'/(read|Identity|Cast(_\d+)?)$')
# op -> {input index: variable kind}
_PARAM_INPUTS = {'Conv2D': {1: 'weights'}, 'BiasAdd': {1: 'biases'},
                 'FusedBatchNorm': {1: 'gamma', 2: 'beta', 3: 'moving_mean', 4: 'moving_variance'}}
_PARAM_INPUTS['FusedBatchNormV2'] = _PARAM_INPUTS['FusedBatchNormV3'] = _PARAM_INPUTS['FusedBatchNorm']


def _attr_string(raw: Optional[bytes]) -> Optional[str]:
    """AttrValue.s (field 2)."""
    if raw is None:
        return None
    for fno, wt, v in _fields(memoryview(raw)):
        if fno == 2 and wt == _LEN:
            return bytes(v).decode()
    return None


def _attr_ints(raw: Optional[bytes]) -> List[int]:
    """AttrValue.list.i (field 1 -> ListValue field 3, packed or repeated)."""
    out: List[int] = []
    if raw is None:
        return out
    for fno, wt, v in _fields(memoryview(raw)):
        if fno == 1 and wt == _LEN:
            for f2, w2, v2 in _fields(v):
                if f2 == 3 and w2 == _LEN:
                    p = 0
                    while p < len(v2):
                        x, p = _read_varint(v2, p)
                        out.append(_signed64(x))
                elif f2 == 3 and w2 == _VARINT:
                    out.append(_signed64(v2))
    return out


def _producer(nodes: Dict[str, 'Node'], ref: str) -> Optional['Node']:
    name = ref[1:] if ref.startswith('^') else ref
    return nodes.get(name.split(':')[0])


def _variable_name(chain: List[str], consumer: 'Node', kind: str) -> str:
    """Slim variable name of a parameter reached through `chain` (node names, constant first)."""
    for name in chain:
        base = _FOLD_RE.sub('', name)
        while _PASS_RE.search(base):
            base = _PASS_RE.sub('', base)
        if base.rsplit('/', 1)[-1] in _KINDS:
            return base
    return consumer.name.rsplit('/', 1)[0] + '/' + kind      # '<layer scope>/Conv2D' -> '<layer scope>/weights'


def structural_params(nodes: Dict[str, 'Node']) -> Dict[str, np.ndarray]:
    """Parameters identified by what consumes them (module docstring)."""
    found: Dict[str, np.ndarray] = {}
    for n in nodes.values():
        for idx, kind in _PARAM_INPUTS.get(n.op, {}).items():
            if idx >= len(n.inputs):
                continue
            chain: List[str] = []
            cur = _producer(nodes, n.inputs[idx])
            hops = 0
            while cur is not None and cur.op in ('Identity', 'Cast', 'StopGradient') and cur.inputs and hops < 8:
                chain.append(cur.name)
                cur = _producer(nodes, cur.inputs[0])
                hops += 1
            if cur is None or cur.op != 'Const' or cur.tensor is None or cur.tensor.dtype not in (np.float32, np.float16):
                continue                                        # an activation (e.g. a 1x1 conv of two tensors): not a parameter
            chain.append(cur.name)
            name = _variable_name(chain[::-1], n, kind)
            t = cur.tensor
            if name in found and not np.array_equal(np.asarray(found[name], np.float32), np.asarray(t, np.float32)):
                raise ValueError(f'two different constants resolve to the variable {name!r}')
            found[name] = t
    return found


def infer_centered_stride(nodes: Dict[str, 'Node'], params: Dict[str, np.ndarray]) -> Optional[bool]:
    """centered_stride of the export, read off the padding modes of the STRIDED 3x3 bottleneck conv2 nodes.

    With centered_stride=True the reference centres exactly ONE block, c[i_last] (resnet_v2.py:278-286,301-309): its
    strided conv2 is a plain SAME convolution, every other strided conv2 stays explicit Pad + VALID
    (resnet_utils.py:120-135).  A default stride-16 export therefore has block1/unit_3 VALID and block2/unit_4 SAME.
    True if ANY strided conv2 is SAME, False if all of them are VALID, None when the graph has no strided unit
    (stride 4: every block runs atrous; the caller then takes the reference default, options.py:118)."""
    seen = False
    for n in nodes.values():
        if n.op != 'Conv2D' or '/bottleneck_v2/conv2' not in n.name:
            continue
        strides = _attr_ints(n.attrs.get('strides'))
        if len(strides) == 4 and max(strides) == 2:
            pad = _attr_string(n.attrs.get('padding'))
            if pad == 'SAME':
                return True
            if pad == 'VALID':
                seen = True
    return False if seen else None


def extract_model(nodes: Dict[str, Node], stride: Optional[int] = None, centered_stride: Optional[bool] = None):
    """centered_stride: None = infer from the graph (default True, options.py:118, when it has no strided unit)."""
    from metro_pose3d_amd.spec import ModelSpec
    params: Dict[str, np.ndarray] = dict(structural_params(nodes))
    arch = None
    for k in params:
        m = _VAR_RE.match(k)
        if m:
            arch = int(m.group(2))
    structural = set(params)
    for name, n in nodes.items():
        if n.op != 'Const' or n.tensor is None:
            continue
        m = _VAR_RE.match(name)
        if not m or n.tensor.dtype not in (np.float32, np.float16):
            continue
        var = m.group(1)
        if var in structural:
            continue                                            # identified by its consumer already
        arch = int(m.group(2))
        exact = m.group(4) is None
        # prefer the original fp32 variable over a folded fp16 copy
        if var not in params or (exact and n.tensor.dtype == np.float32) or \
                (params[var].dtype == np.float16 and n.tensor.dtype == np.float32):
            params[var] = n.tensor
    if arch is None:
        raise ValueError('no MainPart/resnet_v2_{50,101} variables found in the graph')
    params = {k: np.asarray(v, dtype=np.float32) for k, v in params.items()}
    root = f'MainPart/resnet_v2_{arch}'
    if root + '/logits/weights' not in params:
        raise ValueError(f'{root}/logits/weights missing')
    c_head = params[root + '/logits/weights'].shape[3]
    base_width = params[root + '/conv1/weights'].shape[3]

    names = nodes.get('joint_names')
    if names is None or names.tensor is None:
        raise ValueError("graph has no 'joint_names' constant (reference main.py:140)")
    out_names = [b.decode() if isinstance(b, bytes) else str(b) for b in names.tensor.reshape(-1)]

    depth = j_head = side = None
    for n in nodes.values():                                 # Reshape shape of volumetric.py:231
        t = n.tensor
        if n.op == 'Const' and t is not None and t.dtype in (np.int32, np.int64) and t.size == 5:
            v = [int(x) for x in t.reshape(-1)]
            if v[0] == -1 and v[3] == v[4] and v[1] * v[2] == c_head:
                depth, j_head, side = v[1], v[2], v[3]
    if depth is None:
        depth = 8                                            # options.py:113
        j_head = c_head // depth
    if stride is None:
        if side is None:
            raise ValueError('cannot infer the stride (no [-1, depth, J, S, S] reshape constant); pass stride=')
        stride = 256 // side
    from metro_pose3d_amd.joints import skeleton
    dataset = None
    for cand in ('h36m', 'merged', 'many19'):
        sk = skeleton(cand)
        if sk.n_head == j_head and list(sk.names) == out_names:
            dataset = cand
    if dataset is None:
        raise ValueError(f'unrecognised skeleton: {len(out_names)} output joints {out_names[:4]}..., head {j_head}')
    if centered_stride is None:
        inferred = infer_centered_stride(nodes, params)
        centered_stride = True if inferred is None else inferred
    spec = ModelSpec(arch=arch, stride=stride, dataset=dataset, depth=depth, centered_stride=centered_stride,
                     base_width=base_width)
    edges = nodes.get('joint_edges')
    if edges is not None and edges.tensor is not None:
        if not np.array_equal(np.asarray(edges.tensor).reshape(-1, 2), spec.skeleton.edges_array()):
            raise ValueError('joint_edges of the graph differ from the built-in skeleton table')
    return spec, params


def load_frozen_graph(path: str, stride: Optional[int] = None, centered_stride: Optional[bool] = None):
    return extract_model(read_graph(path), stride=stride, centered_stride=centered_stride)


# ---- minimal writer (tests + exporting synthetic models in the reference's container format) --------
def _varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _len_field(fno: int, payload: bytes) -> bytes:
    return _varint((fno << 3) | _LEN) + _varint(len(payload)) + payload


def _var_field(fno: int, v: int) -> bytes:
    return _varint((fno << 3) | _VARINT) + _varint(v)


_DT_OF = {np.dtype(np.float32): DT_FLOAT, np.dtype(np.float16): DT_HALF, np.dtype(np.int32): DT_INT32,
          np.dtype(np.int64): DT_INT64}


def _tensor_proto(arr: np.ndarray) -> bytes:
    if arr.dtype == object:
        shape = b''.join(_len_field(2, _var_field(1, d)) for d in arr.shape)
        return _var_field(1, DT_STRING) + _len_field(2, shape) + b''.join(_len_field(8, bytes(s)) for s in arr.reshape(-1))
    arr = np.ascontiguousarray(arr)
    shape = b''.join(_len_field(2, _var_field(1, d)) for d in arr.shape)
    return _var_field(1, _DT_OF[arr.dtype]) + _len_field(2, shape) + _len_field(4, arr.tobytes())


def _const_node(name: str, arr: np.ndarray) -> bytes:
    attr = _len_field(1, b'value') + _len_field(2, _len_field(8, _tensor_proto(arr)))
    body = _len_field(1, name.encode()) + _len_field(2, b'Const') + _len_field(5, attr)
    return _len_field(1, body)


def write_frozen_graph(path: str, spec, params: Dict[str, np.ndarray], fp16_folded: bool = False) -> None:
    """Writes a GraphDef holding what `extract_model` needs, in the reference's naming.
    fp16_folded=True mimics fold_constants replacing Cast(variable) by an fp16 Const named after it."""
    sk = spec.skeleton
    out = bytearray()
    ph = _len_field(1, b'input') + _len_field(2, b'Placeholder')
    out += _len_field(1, ph)
    for k, v in params.items():
        if fp16_folded and not k.endswith(('gamma', 'beta')):
            out += _const_node(k + '/read/_7__cf__7', np.asarray(v, dtype=np.float16))
        else:
            out += _const_node(k, np.asarray(v, dtype=np.float32))
    side = spec.heatmap_side
    out += _const_node('MainPart/Reshape/shape', np.array([-1, spec.depth, sk.n_head, side, side], dtype=np.int32))
    names = np.empty(sk.n_out, dtype=object)
    names[:] = sk.names_bytes()
    out += _const_node('joint_names', names)
    out += _const_node('joint_edges', sk.edges_array())
    with open(path, 'wb') as f:
        f.write(bytes(out))
