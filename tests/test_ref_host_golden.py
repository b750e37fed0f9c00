"""The product's dispatch compiler against dispatch streams RECORDED from the reference's own host code (tools/make_ref_host_golden.py -> tests/golden/ref_host_dispatches.json):
one digest per dispatch over its name, shader, grid, resource list and constant bytes. Needs neither /root/reference nor oracle/_ref -- this is the form of tests/test_ref_host.py that
travels. The one known difference (REBLUR_DIFFUSE_SPECULAR_SH's transient pool, see tests/test_ref_host.py KNOWN) shows in that denoiser's instance digest and in the grid of one clear."""
import json

import pytest

import ref_host_golden as G

RECORDED = json.load(open(G.PATH))


@pytest.mark.parametrize("key", sorted(RECORDED))
def test_dispatch_stream_equals_the_recorded_reference_host(key):
    name = key.split(" ", 1)[0]
    overrides = json.loads(key.split(" ", 1)[1]) if " " in key else None
    instance, frames = G.stream(name, overrides)
    want_instance, want_frames = RECORDED[key]
    assert [len(f) for f in frames] == [len(f) for f in want_frames]
    differing = [(f, i) for f, (a, b) in enumerate(zip(frames, want_frames)) for i, (x, y) in enumerate(zip(a, b)) if x != y]
    if name == "REBLUR_DIFFUSE_SPECULAR_SH":
        assert instance != want_instance and len(differing) == 1 and differing[0][0] == 0  # the clear of the tile plane on the restart frame (its grid)
    else:
        assert instance == want_instance
        assert not differing, differing[:8]
