"""TensorFlow checkpoint (tensor bundle V2) reader / writer, SURVEY 8f-2.  No TensorFlow here: the writer is pinned to
the format rules by a hand-assembled byte image, the reader by round trips through multi-block tables, and both by the
standard CRC-32C check value."""
import struct

import numpy as np
import pytest
import torch

from scanobjectnn_amd import tf_checkpoint as T


def test_crc32c_check_value_and_mask():
    assert T.crc32c(b"123456789") == 0xE3069283                    # the CRC-32C (Castagnoli) check value
    assert T.crc32c(b"") == 0
    c = T.crc32c(b"foo")
    m = T.masked_crc32c(b"foo")
    rot = (m - 0xa282ead8) & 0xFFFFFFFF                            # leveldb's Unmask: undo the add, rotate back
    assert ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF == c and m != c


def test_writer_matches_hand_assembled_image(tmp_path):
    """one float32 vector 'a' = [1, 2]: every byte of the index derived by hand from the format rules"""
    prefix = str(tmp_path / "model.ckpt")
    T.write_checkpoint(prefix, {"a": np.array([1.0, 2.0], dtype=np.float32)})
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    assert data == struct.pack("<ff", 1.0, 2.0)
    header = bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])           # num_shards=1, version{producer=1}
    entry = bytes([0x08, 0x01, 0x12, 0x04, 0x12, 0x02, 0x08, 0x02, 0x28, 0x08, 0x35]) + \
        struct.pack("<I", T.masked_crc32c(data))                    # dtype=DT_FLOAT, shape{dim{size=2}}, size=8, crc
    block = bytes([0, 0, len(header)]) + header + bytes([0, 1, len(entry)]) + b"a" + entry + \
        struct.pack("<II", 0, 1)                                    # two entries, one restart point at 0
    def framed(b):
        return b + b"\x00" + struct.pack("<I", T.masked_crc32c(b + b"\x00"))
    meta = struct.pack("<II", 0, 1)                                 # empty metaindex block
    handle_data = bytes([0, len(block)])
    index_block = bytes([0, 1, len(handle_data)]) + b"a" + handle_data + struct.pack("<II", 0, 1)
    off_meta = len(block) + 5
    off_index = off_meta + len(meta) + 5
    footer = bytes([off_meta, len(meta), off_index, len(index_block)])
    want = framed(block) + framed(meta) + framed(index_block) + footer + b"\x00" * (40 - len(footer)) + \
        struct.pack("<Q", 0xdb4775248b80fb57)
    assert open(prefix + ".index", "rb").read() == want
    assert np.array_equal(T.read_checkpoint(prefix, verify=True)["a"], [1.0, 2.0])


def test_round_trip_many_blocks_and_prefix_compression(tmp_path):
    rng = np.random.RandomState(0)
    variables = {}
    for l in range(1, 4):
        for c in range(6):
            s = "layer%d/conv%d" % (l, c)
            variables[s + "/weights"] = rng.randn(1, 1, 3 + c, 8).astype(np.float32)
            variables[s + "/biases"] = rng.randn(8).astype(np.float32)
            variables[s + "/bn/moving_mean"] = rng.randn(8).astype(np.float32)
    variables["global_step"] = np.array(7, dtype=np.int64)          # scalar, other dtype
    variables["empty"] = np.zeros((0, 4), dtype=np.float32)
    variables["flags"] = np.array([True, False, True])
    prefix = str(tmp_path / "m")
    T.write_checkpoint(prefix, variables, block_size=200)           # forces a dozen data blocks
    header, entries = T.read_index(prefix + ".index")
    assert header[1] == [1] and set(entries) == set(variables)
    got = T.read_checkpoint(prefix, verify=True)
    for k, v in variables.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "m")
    T.write_checkpoint(prefix, {"w": np.arange(12, dtype=np.float32).reshape(3, 4)})
    raw = bytearray(open(prefix + ".index", "rb").read())
    raw[5] ^= 0x40
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        T.read_index(prefix + ".index")
    raw[5] ^= 0x40
    raw[-1] ^= 0xFF
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="magic"):
        T.read_index(prefix + ".index")
    raw[-1] ^= 0xFF
    open(prefix + ".index", "wb").write(bytes(raw))
    d = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    d[3] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(d))
    assert T.read_checkpoint(prefix)["w"].shape == (3, 4)           # unverified read still works
    with pytest.raises(ValueError, match="tensor checksum"):
        T.read_checkpoint(prefix, verify=True)


def test_name_map():
    assert T.tf_name_to_key("layer1/conv0/weights") == "graph.layer1/conv0/weights"
    assert T.tf_name_to_key("layer1/conv0/bn/moving_variance") == "graph.layer1/conv0/bn/moving_variance"
    assert T.tf_name_to_key("dgcnn1/bn/dgcnn1/bn/moments/Squeeze/ExponentialMovingAverage") == "graph.dgcnn1/bn/moving_mean"
    assert T.tf_name_to_key("transform_net1/tconv1/bn/transform_net1/tconv1/bn/moments/Squeeze_1/ExponentialMovingAverage") \
        == "graph.transform_net1/tconv1/bn/moving_variance"
    for slot in ("fc1/weights/Adam", "fc1/weights/Adam_1", "beta1_power", "beta2_power", "Variable", "global_step"):
        assert T.tf_name_to_key(slot) is None


def test_model_round_trip_through_tf_checkpoint(tmp_path):
    """save a model under the reference's TF variable names, restore it into a differently initialised one"""
    from scanobjectnn_amd.graph import Model
    from scanobjectnn_amd.pointnet import pointnet_cls
    x = torch.zeros(2, 64, 3)
    a = Model(pointnet_cls.get_model, device="cpu", seed=0).build(x)
    b = Model(pointnet_cls.get_model, device="cpu", seed=1).build(x)
    prefix = str(tmp_path / "model.ckpt")
    T.save_tf_checkpoint(a, prefix)
    names = T.read_index(prefix + ".index")[1]
    assert "transform_net1/tconv1/weights" in names and names["transform_net1/tconv1/weights"][1] == (1, 3, 1, 64)
    assert any(not torch.equal(p, q) for p, q in zip(a.state_dict().values(), b.state_dict().values()))
    loaded, missing, unexpected = T.load_tf_checkpoint(b, prefix, verify=True)
    assert not missing and not unexpected and len(loaded) == len(a.state_dict())
    assert all(torch.equal(p, q) for p, q in zip(a.state_dict().values(), b.state_dict().values()))
    # optimizer slots in the file are ignored; a variable the model lacks is an error in strict mode
    extra = {k: v for k, v in T.read_checkpoint(prefix).items()}
    extra["fc1/weights/Adam"] = np.zeros((3,), dtype=np.float32)
    extra["beta1_power"] = np.array(0.9, dtype=np.float32)
    T.write_checkpoint(prefix + "2", extra)
    assert T.load_tf_checkpoint(b, prefix + "2")[2] == []
    extra["not/in/the/model"] = np.zeros((2,), dtype=np.float32)
    T.write_checkpoint(prefix + "3", extra)
    with pytest.raises(KeyError):
        T.load_tf_checkpoint(b, prefix + "3")


from indep_bundle import _indep_bundle, _indep_crc32c, _indep_mask  # noqa: E402  (tests/indep_bundle.py)


def test_reader_against_independently_assembled_bundle(tmp_path):
    rng = np.random.default_rng(3)
    tensors = {
        "layer1/conv0/weights": rng.standard_normal((1, 1, 3, 64)).astype(np.float32),
        "layer1/conv0/biases": rng.standard_normal(64).astype(np.float32),
        "layer1/conv0/bn/beta": rng.standard_normal(64).astype(np.float32),
        "layer1/conv0/bn/gamma": rng.standard_normal(64).astype(np.float32),
        "layer1/conv0/bn/moving_mean": rng.standard_normal(64).astype(np.float32),
        "layer1/conv0/bn/moving_variance": rng.random(64).astype(np.float32),
        "fc3/weights": rng.standard_normal((256, 15)).astype(np.float32),
        "fc3/biases": rng.standard_normal(15).astype(np.float32),
        "global_step": np.array(12345, dtype=np.int64),
        "some/int/table": rng.integers(-5, 5, (4, 3)).astype(np.int32),
    }
    prefix = str(tmp_path / "model.ckpt")
    _indep_bundle(prefix, tensors)
    header, entries = T.read_index(prefix + ".index", verify=True)
    assert header[1][0] == 2 and set(entries) == set(tensors)            # two shards, every variable listed
    got = T.read_checkpoint(prefix, verify=True)
    assert set(got) == set(tensors)
    for name, want in tensors.items():
        assert got[name].dtype == want.dtype and got[name].shape == want.shape and np.array_equal(got[name], want), name
    # the independent CRC agrees with the module's table-driven one (RFC 3720 check value included)
    assert _indep_crc32c(b"123456789") == 0xE3069283 == T.crc32c(b"123456789")
    blob = rng.integers(0, 256, 777, dtype=np.uint8).tobytes()
    assert _indep_mask(_indep_crc32c(blob)) == T.masked_crc32c(blob)
    # a flipped payload byte in shard 1 is caught by the tensor checksum
    p1 = prefix + ".data-00001-of-00002"
    raw = bytearray(open(p1, "rb").read())
    raw[5] ^= 0x40
    open(p1, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        T.read_checkpoint(prefix, verify=True)
