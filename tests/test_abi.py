"""The C-ABI library loads without a GPU and exports every symbol include/vnl.h declares;
compute entry points fail loudly when no HIP device exists (there is no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

import helpers as H
from vnl_brax_imitation_amd import _lib
from vnl_brax_imitation_amd.model import blob

HEADER = os.path.join(H.ROOT, "include", "vnl.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vnl_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_all_exported():
    lib = _lib.load_library()  # built by __graft_entry__.build()
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert set(_lib.EXPORTS) == set(names)
    assert lib.vnl_version() == 1


def test_missing_library_is_a_loud_error(tmp_path):
    with pytest.raises(_lib.VnlError, match="no CPU fallback"):
        _lib.load_library(str(tmp_path / "nope.so"))


def test_bad_blob_and_missing_device_are_reported():
    lib = _lib.load_library()
    h = C.c_void_p()
    bad = C.create_string_buffer(b"garbage" * 10)
    assert lib.vnl_model_create(bad, 70, C.byref(h)) == -2
    assert b"magic" in lib.vnl_last_error()
    good = blob.to_blob(H.model())
    buf = C.create_string_buffer(good, len(good))
    assert lib.vnl_model_create(buf, len(good), C.byref(h)) == 0
    if not torch.cuda.is_available():
        from vnl_brax_imitation_amd.envs.rodent import RodentTracking

        with pytest.raises(_lib.VnlError, match="no CPU fallback|HIP device"):
            RodentTracking(H.reference_clip(), num_envs=2, device="cpu", **H.env_kwargs())
        with pytest.raises(_lib.VnlError, match="no HIP device|hipSetDevice|HIP"):
            with H.backend(lib):
                RodentTracking(H.reference_clip(), num_envs=2, device="cpu", **H.env_kwargs())
    lib.vnl_model_destroy(h)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(H.ROOT, "vnl-brax-imitation_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"import\s+oracle|from\s+oracle|oracle[/.]oracle|liborc|vnl_oracle|orc_", txt), \
                    os.path.join(dp, f)


def test_argument_validation_of_the_helper_entry_points():
    """vnl_rollout_post / vnl_ppo_head / vnl_policy_forward reject malformed descriptors with an error code and a
    message instead of launching (checked on the product library: validation happens before any HIP call)."""
    lib = _lib.load_library()
    d = _lib.PostDesc()
    assert lib.vnl_rollout_post(C.byref(d), 4, None) == -1 and b"null" in lib.vnl_last_error()
    d.done = C.c_void_p(8)  # never dereferenced: validation fails first
    d.num_ops = _lib.POST_MAX_OPS + 1
    assert lib.vnl_rollout_post(C.byref(d), 4, None) == -1 and b"too many" in lib.vnl_last_error()
    d.num_ops = 1  # op 0 has no source
    assert lib.vnl_rollout_post(C.byref(d), 4, None) == -1 and b"bad op" in lib.vnl_last_error()
    assert lib.vnl_rollout_post(None, 4, None) == -1
    a = _lib.PPOHeadArgs()
    work = C.c_void_p(8)
    assert lib.vnl_ppo_head(C.byref(a), work, None) == -1 and b"sizes" in lib.vnl_last_error()
    a.T, a.B, a.act, a.latent = 2, 2, 3, 4
    assert lib.vnl_ppo_head(C.byref(a), work, None) == -1 and b"null buffer" in lib.vnl_last_error()
    assert lib.vnl_policy_forward(None, None, None, None, None, None, None, None, 1, 0, None, None, None, None, None,
                                  None, None, None, None) == -1
