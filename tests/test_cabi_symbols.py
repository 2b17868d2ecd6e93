"""CPU: the C-ABI library builds, loads and exports every symbol include/unislam_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    import unislam_amd
    return unislam_amd


def header_functions(name="unislam_hip.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(us_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(built):
    names = header_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(built.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/unislam_hip.h but not exported"
    from unislam_amd import _lib
    assert sorted(_lib.SIGNATURES.keys()) == names            # the ctypes table covers exactly the header
    assert _lib.lib().us_abi_version() == 2
    # the measured-slower variants live in their own header and are NOT part of the shipped library
    exp = header_functions("unislam_hip_experiments.h")
    assert sorted(_lib.EXPERIMENT_SIGNATURES.keys()) == exp and not set(exp) & set(names)
    assert not _lib.has_experiments() and not any(hasattr(lib, n) for n in exp)


def test_descriptor_and_argument_errors_without_gpu(built):
    us = built
    d = us.make_grid_desc(16, 2, 16, 16, 1.2996847159335432)
    assert d.n_params == 1736800 and list(d.resolution[:16])[-1] == 817
    with pytest.raises(us.UniSlamHipError):
        us.make_grid_desc(40, 2, 16, 16, 1.3)                  # too many levels
    with pytest.raises(us.UniSlamHipError):
        us.make_grid_desc(16, 3, 16, 16, 1.3)                  # unsupported features per level
    import torch
    enc = us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 8,
                                  "base_resolution": 16, "per_level_scale": 1.5})
    with pytest.raises(us.UniSlamHipError):
        enc(torch.rand(5, 3))                                  # CPU tensors are refused: there is no CPU fallback
    from unislam_amd import _lib
    assert _lib.lib().us_hashgrid_fwd(None, None, None, 5, None, None, 0, None) == -1   # NULL descriptor -> US_ERR_NULL
    assert b"NULL" in _lib.lib().us_last_error()


def test_descriptor_matches_oracle(built):
    import unislam_oracle as O
    for res, l2 in [(816, 16), (816, 19), (456, 16), (744, 16)]:
        a = built.make_grid_desc(16, 2, l2, 16, O.per_level_scale(res))
        b = O.make_grid_desc(16, 2, l2, 16, O.per_level_scale(res))
        assert a.n_params == b.n_params
        assert list(a.scale[:16]) == list(b.scale[:16]) and list(a.offset[:17]) == list(b.offset[:17])


def test_modules_pickle_deepcopy_and_state_dict_keys(built):
    import copy
    import pickle
    import torch
    us = built
    cfg = {"grid_mode": "hash_grid", "grid": {"tcnn_network": False}}
    dec = us.Decoders(cfg)
    assert set(dec.state_dict().keys()) == {"beta", "linears.0.weight", "linears.0.bias", "linears.1.weight", "linears.1.bias",
                                            "c_linears.0.weight", "c_linears.0.bias", "c_linears.1.weight", "c_linears.1.bias",
                                            "output_linear.weight", "output_linear.bias", "c_output_linear.weight",
                                            "c_output_linear.bias"}
    d2 = pickle.loads(pickle.dumps(dec)); d3 = copy.deepcopy(dec)
    for a, b, c in zip(dec.parameters(), d2.parameters(), d3.parameters()):
        assert torch.equal(a, b) and torch.equal(a, c)
    enc = us.HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 8,
                                  "base_resolution": 16, "per_level_scale": 1.5})
    e2 = pickle.loads(pickle.dumps(enc)); e3 = copy.deepcopy(enc)
    assert torch.equal(enc.params, e2.params) and torch.equal(enc.params, e3.params) and e2.desc.n_params == enc.desc.n_params
    packed = us.Decoders.pack_linear_params(dec.linears, dec.output_linear)
    assert packed.numel() == 32 * 16 + 16 * 16 + 16 * 16 + 16 + 16 + 16


def test_checkpoint_roundtrip_reference_keys(tmp_path):
    """unislam_amd.checkpoint: the reference's .tar keys (Logger.py:36-46) + the two tables; round trip on the CPU (no kernels)."""
    import types
    import torch
    import unislam_amd as us
    from unislam_amd.checkpoint import load_checkpoint, save_checkpoint
    ecfg = {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 8, "base_resolution": 4, "per_level_scale": 1.5}
    mk = lambda seed: us.HashGridEncoding(3, ecfg, seed=seed)
    cfg = {"grid_mode": "hash_grid", "grid": {"tcnn_network": False}}
    torch.manual_seed(0)
    dec, es, ec = us.Decoders(cfg, c_dim=32), mk(1), mk(2)
    slam = types.SimpleNamespace(decoders=dec, es=es, ec=ec, gt_c2w_list=torch.rand(5, 4, 4), estimate_c2w_list=torch.rand(5, 4, 4),
                                 mapper=types.SimpleNamespace(keyframe_list=[0, 4]))
    p = save_checkpoint(str(tmp_path / "ckpts" / "00004.tar"), slam, 4)
    ck = torch.load(p, map_location="cpu", weights_only=False)
    assert {"decoder_state_dict", "gt_c2w_list", "estimate_c2w_list", "keyframe_list", "idx", "tracking_rendered_weight_list",
            "addtional_map_records"} <= set(ck.keys())
    assert set(ck["decoder_state_dict"].keys()) == set(dec.state_dict().keys())
    torch.manual_seed(1)
    dec2, es2, ec2 = us.Decoders(cfg, c_dim=32), mk(3), mk(4)
    out = load_checkpoint(p, dec2, es2, ec2)
    assert out["idx"] == 4 and out["keyframe_list"] == [0, 4]
    assert all(torch.equal(a, b) for a, b in zip(dec.state_dict().values(), dec2.state_dict().values()))
    assert torch.equal(es.params, es2.params) and torch.equal(ec.params, ec2.params)
