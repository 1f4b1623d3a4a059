"""Row f1: TF-free frozen-GraphDef (.pb) reader.  No TensorFlow and no released .pb exist offline, so
the reader is tested against files produced by the in-repo writer (same protobuf wire format and the
reference's node naming), including the fp16 constant-folded variant."""
import numpy as np
import pytest

from metro_pose3d_amd import ModelSpec, load_model, synth
from metro_pose3d_amd import tfgraph


@pytest.mark.parametrize('spec', [ModelSpec(50, 16, 'h36m', base_width=8), ModelSpec(101, 8, 'merged', base_width=8),
                                  ModelSpec(50, 4, 'many19', base_width=8)],
                         ids=lambda s: f'rn{s.arch}-s{s.stride}-{s.dataset}')
def test_pb_roundtrip_fp32(tmp_path, spec):
    params = synth.make_params(spec.arch, spec.n_head_channels, spec.base_width, seed=2)
    path = str(tmp_path / 'model.pb')
    tfgraph.write_frozen_graph(path, spec, params)
    spec2, params2 = load_model(path)                       # dispatches on content, not extension
    assert spec2 == spec
    assert sorted(params2) == sorted(params)
    assert all(np.array_equal(params[k], params2[k]) for k in params)


def test_pb_fp16_folded_constants(tmp_path):
    spec = ModelSpec(50, 32, 'h36m', base_width=8)
    params = synth.make_params(50, spec.n_head_channels, 8, seed=3)
    path = str(tmp_path / 'model_fp16.pb')
    tfgraph.write_frozen_graph(path, spec, params, fp16_folded=True)
    spec2, params2 = load_model(path)
    assert spec2 == spec
    for k, v in params.items():
        exp = v if k.endswith(('gamma', 'beta')) else v.astype(np.float16).astype(np.float32)
        assert params2[k].dtype == np.float32 and np.array_equal(params2[k], exp), k


def test_pb_node_and_tensor_decoding(tmp_path):
    spec = ModelSpec(50, 16, 'h36m', base_width=8)
    params = synth.make_params(50, spec.n_head_channels, 8)
    path = str(tmp_path / 'm.pb')
    tfgraph.write_frozen_graph(path, spec, params)
    nodes = tfgraph.read_graph(path)
    assert nodes['input'].op == 'Placeholder'
    assert [b.decode() for b in nodes['joint_names'].tensor][:3] == ['pelv', 'rhip', 'rkne']
    assert nodes['joint_edges'].tensor.dtype == np.int64 and nodes['joint_edges'].tensor.shape == (16, 2)
    assert nodes['MainPart/Reshape/shape'].tensor.tolist() == [-1, 8, 17, 16, 16]
    # splat-encoded and value-list tensors (float_val / int_val paths of TensorProto)
    body = (tfgraph._var_field(1, tfgraph.DT_FLOAT) + tfgraph._len_field(2, tfgraph._len_field(2, tfgraph._var_field(1, 4))) +
            tfgraph._len_field(5, np.float32([2.5]).tobytes()))
    assert tfgraph._parse_tensor(memoryview(body)).tolist() == [2.5] * 4
    body = (tfgraph._var_field(1, tfgraph.DT_INT32) + tfgraph._len_field(2, tfgraph._len_field(2, tfgraph._var_field(1, 3))) +
            tfgraph._len_field(7, b''.join(tfgraph._varint(v) for v in (-1, 8, 300))))
    assert tfgraph._parse_tensor(memoryview(body)).tolist() == [-1, 8, 300]


def test_pb_errors(tmp_path):
    bad = tmp_path / 'bad.pb'
    bad.write_bytes(b'\x00\x01garbage')
    with pytest.raises(ValueError):
        load_model(str(bad))
    spec = ModelSpec(50, 16, 'h36m', base_width=8)
    params = synth.make_params(50, spec.n_head_channels, 8)
    params.pop('MainPart/resnet_v2_50/logits/weights')
    p = str(tmp_path / 'nologits.pb')
    tfgraph.write_frozen_graph(p, spec, params)
    with pytest.raises(ValueError, match='logits'):
        load_model(p)


@pytest.mark.gpu
def test_estimate_pose_from_pb(cuda, tmp_path):
    import torch
    from metro_pose3d_amd import save_model
    from metro_pose3d_amd.inference import estimate_pose
    spec = ModelSpec(50, 16, 'h36m', base_width=16)
    params = synth.make_params(50, spec.n_head_channels, 16, seed=4, logit_gain=0.8)
    images = synth.make_images(2)
    pb, npz = str(tmp_path / 'm.pb'), str(tmp_path / 'm.npz')
    tfgraph.write_frozen_graph(pb, spec, params)
    save_model(npz, spec, params)
    a, ea, na = estimate_pose(images, pb, precision='f64')
    b, eb, nb = estimate_pose(images, npz, precision='f64')
    assert torch.equal(a, b) and np.array_equal(ea, eb) and list(na) == list(nb)


def _official_protobuf_classes():
    """GraphDef / NodeDef / AttrValue / TensorProto / TensorShapeProto declared with the official
    protobuf runtime (field numbers of tensorflow/core/framework/*.proto), to encode independently of
    the in-repo writer."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name='tfmini.proto', package='tfmini', syntax='proto3')

    def msg(name, fields, nested=None):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = tname
        return m

    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    dim = msg('Dim', [('size', 1, F.TYPE_INT64, OPT, None), ('name', 2, F.TYPE_STRING, OPT, None)])
    msg('TensorShapeProto', [('dim', 2, F.TYPE_MESSAGE, REP, '.tfmini.Dim')])
    msg('TensorProto', [('dtype', 1, F.TYPE_INT32, OPT, None), ('tensor_shape', 2, F.TYPE_MESSAGE, OPT, '.tfmini.TensorShapeProto'),
                        ('tensor_content', 4, F.TYPE_BYTES, OPT, None), ('float_val', 5, F.TYPE_FLOAT, REP, None),
                        ('int_val', 7, F.TYPE_INT32, REP, None), ('string_val', 8, F.TYPE_BYTES, REP, None),
                        ('int64_val', 10, F.TYPE_INT64, REP, None), ('half_val', 13, F.TYPE_INT32, REP, None)])
    msg('ListValue', [('i', 3, F.TYPE_INT64, REP, None)])
    msg('AttrValue', [('list', 1, F.TYPE_MESSAGE, OPT, '.tfmini.ListValue'), ('s', 2, F.TYPE_BYTES, OPT, None),
                      ('i', 3, F.TYPE_INT64, OPT, None), ('type', 6, F.TYPE_INT32, OPT, None),
                      ('tensor', 8, F.TYPE_MESSAGE, OPT, '.tfmini.TensorProto')])
    msg('AttrEntry', [('key', 1, F.TYPE_STRING, OPT, None), ('value', 2, F.TYPE_MESSAGE, OPT, '.tfmini.AttrValue')])
    msg('NodeDef', [('name', 1, F.TYPE_STRING, OPT, None), ('op', 2, F.TYPE_STRING, OPT, None), ('input', 3, F.TYPE_STRING, REP, None),
                    ('device', 4, F.TYPE_STRING, OPT, None), ('attr', 5, F.TYPE_MESSAGE, REP, '.tfmini.AttrEntry')])
    msg('GraphDef', [('node', 1, F.TYPE_MESSAGE, REP, '.tfmini.NodeDef')])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName('tfmini.' + n))
    return get('GraphDef'), get('TensorProto')


def test_reader_against_official_protobuf_encoder(tmp_path):
    """Independent check of the hand-written wire-format decoder: the graph is ENCODED by the
    official protobuf runtime (no in-repo writer involved) with a mix of tensor encodings."""
    GraphDef, _ = _official_protobuf_classes()
    spec = ModelSpec(50, 16, 'h36m', base_width=8)
    params = synth.make_params(50, spec.n_head_channels, 8, seed=9)
    g = GraphDef()
    n = g.node.add(name='input', op='Placeholder')
    n.attr.add(key='dtype').value.type = 1
    for i, (k, v) in enumerate(sorted(params.items())):
        node = g.node.add(name=k, op='Const', device='/device:CPU:0')
        node.attr.add(key='dtype').value.type = 1
        t = node.attr.add(key='value').value.tensor
        t.dtype = 1
        for d in v.shape:
            t.tensor_shape.dim.add(size=d)
        if i % 3 == 0 and v.size <= 4096:
            t.float_val.extend(v.reshape(-1).tolist())        # repeated-field encoding
        else:
            t.tensor_content = v.tobytes()
        # a consumer, so that non-Const nodes with inputs are exercised too
        g.node.add(name=k + '/read', op='Identity', input=[k])
    shp = g.node.add(name='some/Reshape/shape', op='Const').attr.add(key='value').value.tensor
    shp.dtype = 3
    shp.tensor_shape.dim.add(size=5)
    shp.int_val.extend([-1, 8, 17, 16, 16])
    jn = g.node.add(name='joint_names', op='Const').attr.add(key='value').value.tensor
    jn.dtype = 7
    jn.tensor_shape.dim.add(size=17)
    jn.string_val.extend(spec.skeleton.names_bytes())
    je = g.node.add(name='joint_edges', op='Const').attr.add(key='value').value.tensor
    je.dtype = 9
    je.tensor_shape.dim.add(size=16)
    je.tensor_shape.dim.add(size=2)
    je.int64_val.extend(spec.skeleton.edges_array().reshape(-1).tolist())
    path = str(tmp_path / 'official.pb')
    with open(path, 'wb') as f:
        f.write(g.SerializeToString())
    spec2, params2 = load_model(path)
    assert spec2 == spec
    assert all(np.array_equal(params[k], params2[k]) for k in params)


# ---- graphs shaped like a TF 1.13 export of this model (reference main.py:143-161) -------------------------------------
def _tf_like_graph(spec, params, fp16_folded, drop_names=False):
    """variable Const -> Identity '<var>/read' -> Cast -> Conv2D / BiasAdd, BN parameters -> FusedBatchNorm, encoded with
    the official protobuf runtime.  fp16_folded: what `fold_constants` + `strip_unused_nodes` leave of the default fp16
    export -- Cast(read(variable)) of every TRAINABLE variable cast at use (tfu.py:426-440: conv kernels and biases;
    BN statistics and gamma/beta are requested in fp32 by the fused batch norm) replaced by ONE fp16 Const named after
    the Cast node, the fp32 original pruned.  drop_names: the folded constants get names that do not contain the layer
    scope at all (only the consumer identifies them)."""
    from oracle.spec import schedule
    from tests import helpers as H
    GraphDef, _ = _official_protobuf_classes()
    g = GraphDef()
    g.node.add(name='input', op='Placeholder')
    counter = [0]

    def const(name, arr, dtype):
        node = g.node.add(name=name, op='Const')
        t = node.attr.add(key='value').value.tensor
        t.dtype = {np.float32: 1, np.float16: 19}[dtype]
        for d in arr.shape:
            t.tensor_shape.dim.add(size=d)
        t.tensor_content = np.ascontiguousarray(arr.astype(dtype)).tobytes()
        return name

    def variable(var, cast_name, trainable_cast):
        """returns the name the consumer reads"""
        v = params[var]
        if fp16_folded and trainable_cast:
            counter[0] += 1
            k = counter[0]
            name = (f'ConstantFolding/c{k}' if drop_names else f'{cast_name}/_{k}__cf__{k}')
            return const(name, v, np.float16)
        const(var, v, np.float32)
        g.node.add(name=var + '/read', op='Identity', input=[var])
        if trainable_cast:                       # fp16 graph before folding: Cast(read)
            g.node.add(name=cast_name, op='Cast', input=[var + '/read'])
            return cast_name
        return var + '/read'

    def conv(scope, x, stride=1, padding='SAME', bias=False):
        n = g.node.add(name=scope + '/Conv2D', op='Conv2D', input=[x, variable(scope + '/weights', scope + '/Cast', True)])
        n.attr.add(key='strides').value.list.i.extend([1, 1, stride, stride])          # NCHW (options.py:92)
        n.attr.add(key='padding').value.s = padding.encode()
        out = scope + '/Conv2D'
        if bias:
            g.node.add(name=scope + '/BiasAdd', op='BiasAdd', input=[out, variable(scope + '/biases', scope + '/Cast_1', True)])
            out = scope + '/BiasAdd'
        return out

    def bn(scope, x):
        ins = [x] + [variable(f'{scope}/{k}', '', False) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
        g.node.add(name=scope + '/FusedBatchNorm', op='FusedBatchNormV2', input=ins)
        g.node.add(name=scope + '/Relu', op='Relu', input=[scope + '/FusedBatchNorm'])
        return scope + '/Relu'

    root = f'MainPart/resnet_v2_{spec.arch}'
    x = conv(root + '/conv1', 'input', 2, 'VALID', bias=True)
    for u in schedule(H.oracle_spec(spec)):
        sc = f'{root}/{u.name}/bottleneck_v2'
        pre = bn(sc + '/preact', x)
        short = conv(sc + '/shortcut', pre, u.stride, 'SAME', bias=True) if u.c_in != u.c_out else x
        r = bn(sc + '/conv1/BatchNorm', conv(sc + '/conv1', pre))
        # conv2d_same: explicit Pad + VALID (resnet_utils.py:125-135) unless THIS unit is the centred one: only the last
        # unit of block c[i_last] is (resnet_v2.py:278-286), e.g. block2 of a stride-16 net -- block1 stays Pad + VALID
        if u.stride == 2 and not u.centered:
            g.node.add(name=sc + '/Pad', op='Pad', input=[r])
            r = conv(sc + '/conv2', sc + '/Pad', 2, 'VALID')
        else:
            r = conv(sc + '/conv2', r, u.stride, 'SAME')
        r = bn(sc + '/conv2/BatchNorm', r)
        r = conv(sc + '/conv3', r, bias=True)
        g.node.add(name=sc + '/add', op='Add', input=[short, r])
        x = sc + '/add'
    x = bn(root + '/postnorm', x)
    conv(root + '/logits', x, bias=True)
    shp = g.node.add(name='MainPart/Reshape/shape', op='Const').attr.add(key='value').value.tensor
    shp.dtype = 3
    shp.tensor_shape.dim.add(size=5)
    shp.int_val.extend([-1, spec.depth, spec.skeleton.n_head, spec.heatmap_side, spec.heatmap_side])
    jn = g.node.add(name='joint_names', op='Const').attr.add(key='value').value.tensor
    jn.dtype = 7
    jn.tensor_shape.dim.add(size=spec.skeleton.n_out)
    jn.string_val.extend(spec.skeleton.names_bytes())
    je = g.node.add(name='joint_edges', op='Const').attr.add(key='value').value.tensor
    je.dtype = 9
    e = spec.skeleton.edges_array()
    je.tensor_shape.dim.add(size=e.shape[0])
    je.tensor_shape.dim.add(size=2)
    je.int64_val.extend(e.reshape(-1).tolist())
    return g.SerializeToString()


def _expected_params(params, fp16_folded):
    out = {}
    for k, v in params.items():
        cast = fp16_folded and k.endswith(('/weights', '/biases'))
        out[k] = v.astype(np.float16).astype(np.float32) if cast else v.astype(np.float32)
    return out


@pytest.mark.parametrize('fp16_folded,drop_names', [(False, False), (True, False), (True, True)],
                         ids=['fp32-unfolded', 'fp16-folded-cast-names', 'fp16-folded-anonymous'])
@pytest.mark.parametrize('centered', [True, False], ids=['centered', 'not-centered'])
@pytest.mark.parametrize('stride', [16, 32], ids=['s16', 's32'])
def test_tf_export_shaped_graph_is_read_by_structure(tmp_path, fp16_folded, drop_names, centered, stride):
    """The default export is fp16: after fold_constants the kernel of a conv is a Const named after the CAST node
    ('.../conv1/Cast/_7__cf__7'), or not after the layer at all -- the reader finds it through the Conv2D that consumes
    it, and reads centered_stride off the strided conv2 nodes' padding modes: a centred export has ONE SAME-padded strided
    conv2 (block2 at stride 16, block3 at stride 32) behind Pad + VALID ones (block1; blocks 1-2)."""
    spec = ModelSpec(50, stride, 'h36m', base_width=8, centered_stride=centered)
    params = synth.make_params(50, spec.n_head_channels, 8, seed=11)
    path = tmp_path / 'export.pb'
    path.write_bytes(_tf_like_graph(spec, params, fp16_folded, drop_names))
    spec2, params2 = load_model(str(path))
    assert spec2 == spec and spec2.centered_stride == centered
    exp = _expected_params(params, fp16_folded)
    assert sorted(params2) == sorted(exp)
    for k in exp:
        assert params2[k].dtype == np.float32 and np.array_equal(params2[k], exp[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize('fp16_folded', [False, True], ids=['fp32', 'fp16-folded'])
def test_hip_path_from_tf_export_shaped_graph_matches_oracle(cuda, tmp_path, fp16_folded):
    """.pb shaped like the reference's export -> HIP path, against the ORACLE on the variables the graph holds."""
    import torch
    from metro_pose3d_amd.inference import estimate_pose
    from oracle import forward as OF
    from tests import helpers as H
    spec = ModelSpec(50, 16, 'h36m', base_width=16, centered_stride=False)
    params = synth.make_params(50, spec.n_head_channels, 16, seed=4, logit_gain=0.8)
    path = tmp_path / 'export.pb'
    path.write_bytes(_tf_like_graph(spec, params, fp16_folded))
    images = synth.make_images(2)
    poses, _, names = estimate_pose(images, str(path), precision='f64')
    ref = OF.forward(H.oracle_spec(spec), _expected_params(params, fp16_folded), images, torch.float64).numpy()
    assert names[0] == b'pelv'
    assert np.abs(poses.cpu().numpy() - ref).max() <= 1e-3
