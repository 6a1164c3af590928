"""pyn_mesh_box: a rank's block of the reference's box mesh (src/domain/dmplex.py:16-21, 42-95) generated on the device must be the
mesh the host-side construction (DMPlexDom.conn / .xyz, checked against the reference's fixtures in tests/test_domain_cpu.py) uploads,
entry by entry, and must be recognised as the same topology."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pynama_amd import _lib
    return _lib


def box_ctx(lib, dom, rank, size):
    ctx = lib.Context(0)
    if size > 1:
        ctx.comm_init(rank, size, None)                      # detached
        ctx.halo_set(*dom._halo_plan())
    k0, k1 = dom._layers
    ctx.mesh_box(dom.dim, dom.ngl, dom.nelem[:-1] + [k1 - k0], k0, dom.lattice, dom._loc, dom._local_plane_ids(), dom._axes())
    return ctx


def host_ctx(lib, dom, rank, size):
    ctx = lib.Context(0)
    if size > 1:
        ctx.comm_init(rank, size, None)
        ctx.halo_set(*dom._halo_plan())
    ctx.mesh_set(dom.dim, dom.conn, dom.xyz)
    return ctx


@pytest.mark.parametrize("nelem,ngl,size", [([5, 4], 2, 1), ([5, 4], 3, 1), ([3, 4], 5, 1), ([4, 3, 5], 2, 1), ([3, 2, 4], 3, 1),
                                            ([2, 2, 2], 4, 1), ([5, 8], 3, 2), ([4, 9], 2, 3), ([3, 2, 6], 2, 2), ([2, 2, 7], 3, 3)])
def test_generated_mesh_equals_the_host_construction(lib, nelem, ngl, size):
    from pynama_amd.common.comm import Comm
    from pynama_amd.domain.dmplex import DMPlexDom
    dim = len(nelem)
    lo, up = [0.25, -1.0, 0.0][:dim], [1.0, 0.8, 1.7][:dim]
    for r in range(size):
        dom = DMPlexDom(boxMesh={'nelem': nelem, 'lower': lo, 'upper': up}, comm=Comm(r, size))
        dom.setFemIndexing(ngl)
        ctx = box_ctx(lib, dom, r, size)
        conn, xyz = ctx.mesh_get()
        assert conn.shape == dom.conn.shape and np.array_equal(conn, dom.conn)
        assert np.array_equal(xyz, dom.xyz)                            # the same doubles: both pick lattice lines
        ref = host_ctx(lib, dom, r, size)
        assert ctx.mesh_topology() == ref.mesh_topology()
        assert (ctx.n_elem, ctx.n_node, ctx.nn) == (ref.n_elem, ref.n_node, ref.nn)
        ref.close()
        ctx.close()


def test_domain_uses_the_generator_and_builds_host_arrays_only_on_request(lib):
    from pynama_amd.domain.dmplex import DMPlexDom
    dom = DMPlexDom(nelem=[6, 5, 4], lower=[0, 0, 0], upper=[1, 1, 1])
    dom.setFemIndexing(2)
    ctx = dom.ctx
    assert dom._conn is None and dom._xyz is None                      # nothing was built on the host
    assert ctx.mesh_topology() == ("lattice", 7, 6, 5)
    bm = dom.boundaryMaskLocal()
    X = dom.getNodesCoordinates(nodes=[0, 7 * 6 * 5 - 1, 9])
    assert dom._xyz is None
    conn, xyz = ctx.mesh_get()
    assert np.array_equal(conn, dom.conn) and np.array_equal(xyz, dom.xyz)
    assert np.array_equal(X, dom.xyz[[0, 7 * 6 * 5 - 1, 9]])
    on = np.zeros(dom.nLocal, bool)
    for d in range(3):
        on |= (dom.xyz[:, d] == 0.0) | (dom.xyz[:, d] == 1.0)
    assert np.array_equal(bm.astype(bool), on)
    # a jittered mesh has no closed form: it is uploaded
    dj = DMPlexDom(nelem=[4, 4, 4], lower=[0, 0, 0], upper=[1, 1, 1], jitter=0.2)
    dj.setFemIndexing(2)
    c2, x2 = dj.ctx.mesh_get()
    assert dj._xyz is not None and np.array_equal(x2, dj.xyz) and np.array_equal(c2, dj.conn)


def test_bad_arguments_are_refused(lib):
    from pynama_amd.domain.dmplex import DMPlexDom
    dom = DMPlexDom(nelem=[3, 3], lower=[0, 0], upper=[1, 1])
    dom.setFemIndexing(3)
    ctx = lib.Context(0)
    planes = dom._local_plane_ids()
    with pytest.raises(lib.PynamaHipError, match="permutation"):
        ctx.mesh_box(2, 3, [3, 3], 0, dom.lattice, dom._loc, planes[:-1] + [planes[0]], dom._axes())
    with pytest.raises(lib.PynamaHipError, match="planes"):
        ctx.mesh_box(2, 3, [3, 2], 0, dom.lattice, dom._loc, planes, dom._axes())
    with pytest.raises(lib.PynamaHipError, match="outside the element"):
        ctx.mesh_box(2, 3, [3, 3], 0, dom.lattice, dom._loc + 1, planes, dom._axes())
    # an uploaded connectivity with an entry outside the mesh is still refused, now by the device-side check
    conn = dom.conn.copy()
    conn[4, 2] = dom.nLocal
    with pytest.raises(lib.PynamaHipError, match=r"conn\[38\]=49 out of range"):
        ctx.mesh_set(2, conn, dom.xyz)
    ctx.close()
