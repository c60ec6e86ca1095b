"""Reference-source pins that travel: tests/golden/ref_full_sha256.json holds the SHA-256 of what the reference's OWN
kernel source (oracle/_ref/libispc_texcomp_ref_full.so, built in the container from /root/reference) emits for every golden
input x format x preset -- 46 streams (tools/make_ref_full_sha256.py).  The oracle must reproduce every one of them on any
box, with or without the reference tree and the git-ignored oracle/_ref binaries; where the binaries are present they must
reproduce the file too (so a stale file cannot go unnoticed)."""
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_pins():
    with open(os.path.join(ROOT, "tests", "golden", "ref_full_sha256.json")) as f:
        return json.load(f)


def split_key(key):
    parts = key.split(".")
    return parts[0], parts[1], (parts[2] if len(parts) > 2 else None)


def test_inputs_are_the_pinned_inputs(golden_inputs):
    pins = load_pins()
    assert set(pins["_inputs"]) == set(golden_inputs)
    for name, img in golden_inputs.items():
        assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == pins["_inputs"][name], name


def test_oracle_reproduces_every_reference_source_stream(oracle, golden_inputs):
    pins = load_pins()
    assert len(pins["streams"]) == 46
    for key, want in pins["streams"].items():
        name, fmt, prof = split_key(key)
        got = hashlib.sha256(oracle.encode_mt(fmt, golden_inputs[name], prof).tobytes()).hexdigest()
        assert got == want, f"oracle differs from the reference-source stream {key}"


def test_reference_source_build_reproduces_the_file(golden_inputs):
    from oracle import pyref
    if not pyref.available():
        if os.path.exists("/root/reference/IntelCompressionPlugin/kernel.ispc"):
            pytest.fail("the reference tree is here but oracle/_ref/libispc_texcomp_ref_full.so is not built: run build() / make -C oracle/ref_build")
        pytest.skip("no reference tree and no prebuilt oracle/_ref on this box: the committed pins stand in (checked above)")
    pins = load_pins()
    for key, want in pins["streams"].items():
        name, fmt, prof = split_key(key)
        got = hashlib.sha256(pyref.encode_mt(fmt, golden_inputs[name], prof).tobytes()).hexdigest()
        assert got == want, f"{key}: regenerate tests/golden/ref_full_sha256.json (tools/make_ref_full_sha256.py)"
    if os.path.exists("/root/reference/IntelCompressionPlugin/kernel.ispc"):
        with open("/root/reference/IntelCompressionPlugin/kernel.ispc", "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest() == pins["_reference_sources"]["kernel.ispc"]
