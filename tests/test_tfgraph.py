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
    msg('AttrValue', [('s', 2, F.TYPE_BYTES, OPT, None), ('i', 3, F.TYPE_INT64, OPT, None), ('type', 6, F.TYPE_INT32, OPT, None),
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
