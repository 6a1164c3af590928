"""Wall bookkeeping of the no-slip split: the known answers of the reference's own suite
(/root/reference/src/tests/test_nswalls.py:6-373) on pynama_amd.common.nswalls."""
import numpy as np

from pynama_amd.common.nswalls import NoSlipWalls


def _walls(dim):
    lower = np.random.rand(dim)
    return NoSlipWalls(lower=lower, upper=np.random.rand(dim) + lower)


def test_walls_with_velocity():                    # test_nswalls.py:7-30
    ns = _walls(2)
    ns.setWallVelocity(name="up", vel=[1, 0])
    ns.setWallVelocity(name="down", vel=[2, 0])
    assert "up" in ns.getWallsWithVelocity() and "down" in ns.getWallsWithVelocity()
    assert "left" not in ns.getWallsWithVelocity() and "right" not in ns.getWallsWithVelocity()
    ns = _walls(2)
    ns.setWallVelocity(name="left", vel=[0, 1])
    ns.setWallVelocity(name="right", vel=[0, 4])
    assert set(ns.getWallsWithVelocity()) == {"left", "right"}


def test_ignore_velocity_normal():                 # test_nswalls.py:32-47
    ud, lr = _walls(2), _walls(2)
    lr.setWallVelocity(name="left", vel=[1, 0])
    lr.setWallVelocity(name="right", vel=[1, 0])
    ud.setWallVelocity(name="up", vel=[0, 1])
    ud.setWallVelocity(name="down", vel=[0, 1])
    assert not ud.getWallsWithVelocity() and not lr.getWallsWithVelocity()


def test_velocities_and_dofs():                    # test_nswalls.py:49-87
    ns = _walls(2)
    ns.setWallVelocity(name="right", vel=[0, 4])
    vel, dofs = ns.getWallVelocity("right")
    assert vel[0] == 4 and len(vel) == 1 and dofs[0] == 1
    ns.setWallVelocity(name="up", vel=[3, 0])
    vel, dofs = ns.getWallVelocity("up")
    assert vel[0] == 3 and dofs[0] == 0


def test_normals_and_static_dofs_2d():             # test_nswalls.py:89-135
    ns = _walls(2)
    for w in ("left", "right"):
        assert ns.getWalletNormalBySideName(w) == 0 and ns.getStaticDofsByName(w) == [1]
    for w in ("up", "down"):
        assert ns.getWalletNormalBySideName(w) == 1 and ns.getStaticDofsByName(w) == [0]
    st = ns.getStaticDofsByName("up")
    ns.setWallVelocity("up", [5, 0])
    assert len(st) == 0                            # same list object, emptied in place
    assert list(ns.getWallsNames()) == ["left", "right", "up", "down"]


def test_walls_3d():                               # test_nswalls.py 3D cases
    ns = _walls(3)
    assert list(ns.getWallsNames()) == ["left", "right", "up", "down", "back", "front"]
    assert [ns.getWalletNormalBySideName(w) for w in ("left", "up", "front", "back")] == [0, 1, 2, 2]
    assert ns.getStaticDofsByName("front") == [0, 1] and ns.getStaticDofsByName("left") == [1, 2]
    ns.setWallVelocity("up", [3, 0, 4])
    vel, dofs = ns.getWallVelocity("up")
    assert list(vel) == [3, 4] and dofs == [0, 2] and ns.getStaticDofsByName("up") == []
    ns.setWallVelocity("down", [3, 7, 0])          # y is the normal of 'down': ignored
    vel, dofs = ns.getWallVelocity("down")
    assert list(vel) == [3] and dofs == [0] and ns.getStaticDofsByName("down") == [2]
    assert set(ns.getStaticWalls()) == {"left", "right", "back", "front"}


def test_exclude_walls():
    ns = NoSlipWalls([0, 0], [1, 1], exclude=["left", "right"])
    assert list(ns.getWallsNames()) == ["up", "down"] and len(ns) == 2
