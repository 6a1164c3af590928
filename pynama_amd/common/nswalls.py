"""No-slip wall bookkeeping of a box domain (mirrors ``src/common/nswalls.py``: ``NoSlipWalls``
:5-166, ``Wall`` :168-268): wall names, outward-normal axis, which velocity DOFs of a wall are held
at zero ("static") and which carry a prescribed tangential velocity."""
import numpy as np

_NORMAL_AXIS = {"left": 0, "right": 0, "up": 1, "down": 1, "back": 2, "front": 2}


class Wall:
    def __init__(self, num, name, dim):
        self.num, self.name, self.dim = num, name, dim
        self.normal = _NORMAL_AXIS[name]
        self.staticDofs = [d for d in range(dim) if d != self.normal]      # tangential DOFs, all at rest
        self.velocity = None
        self.velDofs = None

    def setWallName(self, name):
        self.name = name

    def getWallName(self):
        return self.name

    def getWallNum(self):
        return self.num

    def computeNormal(self):
        """axis of the wall normal: 0 x, 1 y, 2 z"""
        return self.normal

    def setWallVelocity(self, vel):
        """move the tangential DOFs with a non-zero prescribed value from 'static' to 'moving'; the
        normal component of `vel` is ignored; all-zero tangential velocity is an error (nswalls.py:198-213)"""
        vels, dofs = [], []
        for dof in list(self.staticDofs):
            if vel[dof] != 0:
                vels.append(vel[dof])
                dofs.append(dof)
                self.staticDofs.remove(dof)
        if not dofs:
            raise Exception("Velocity not valid")
        self.velocity = np.array(vels)
        self.velDofs = dofs

    def getWallVelocity(self):
        if self.velocity is not None:
            return self.velocity, self.velDofs
        return [0] * len(self.staticDofs), self.staticDofs

    def getStaticDofs(self):
        return self.staticDofs


class NoSlipWalls:
    def __init__(self, lower, upper, exclude=[]):
        self.dim = len(lower)
        self.lower, self.upper = lower, upper
        sides = ["left", "right", "up", "down"] if self.dim == 2 else ["left", "right", "up", "down", "back", "front"]
        self.walls = {}
        for num, side in enumerate(sides):
            if side not in exclude:
                self.walls[side] = Wall(num, side, self.dim)
        self.staticWalls = list(self.walls.keys())
        self.wallsWithVelocity = []
        self.normals = {name: w.computeNormal() for name, w in self.walls.items()}

    def __iter__(self):
        return iter(self.walls.values())

    def __len__(self):
        return len(self.walls)

    def getWallsNames(self):
        return self.walls.keys()

    def getWallsWithVelocity(self):
        return self.wallsWithVelocity

    def getWallBySideName(self, name):
        return self.walls[name]

    def getStaticWalls(self):
        return self.staticWalls

    def setWallVelocity(self, name, vel):
        try:
            assert len(self.lower) == len(vel)
            self.walls[name].setWallVelocity(vel)
            self.wallsWithVelocity.append(name)
            self.staticWalls.remove(name)
        except Exception:
            return None

    def getWallVelocity(self, name):
        return self.walls[name].getWallVelocity()

    def getStaticDofsByName(self, name):
        return self.walls[name].getStaticDofs()

    def getWalletNormalBySideName(self, name):
        return self.normals[name]
