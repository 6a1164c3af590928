"""Analytic-field cases (mirrors src/cases/custom_func.py:14-83 and the Taylor-Green fields
:173-193): boundary velocity and interior vorticity are taken from closed-form functions."""
from math import cos, exp, pi, sin

from pynama_amd.cases.base_problem import FreeSlip


class CustomFuncCase(FreeSlip):
    def setUp(self):
        self.setUpGeneral()
        self.setUpBoundaryConditions()
        self.setUpEmptyMats()
        self.buildKLEMats()
        self.buildOperators()
        self.nu = self.mu / self.rho
        if self.case == 'taylor-green':
            if self.dim == 2:
                self.velFunction = self.taylorGreenVel_2D
                self.vortFunction = self.taylorGreenVort_2D
            else:
                raise Exception("3D Taylor-Green fields are not part of the pinned path")
        else:
            raise Exception("Case not found")

    def computeInitialCondition(self, startTime):
        allNodes = self.dom.getAllNodes()
        fvort = lambda coords: self.vortFunction(coords, self.nu, t=startTime)
        self.vort = self.dom.applyFunctionVecToVec(allNodes, fvort, self.vort, self.dim_w)

    def generateExactVecs(self, time):
        exactVel = self.mat.K.createVecRight()
        exactVort = self.mat.Rw.createVecRight()
        allNodes = self.dom.getAllNodes()
        fvel = lambda coords: self.velFunction(coords, self.nu, t=time)
        fvort = lambda coords: self.vortFunction(coords, self.nu, t=time)
        exactVel = self.dom.applyFunctionVecToVec(allNodes, fvel, exactVel, self.dim)
        exactVort = self.dom.applyFunctionVecToVec(allNodes, fvort, exactVort, self.dim_w)
        return exactVel, exactVort

    def applyBoundaryConditions(self, time):
        self.vel.set(0.0)
        fvel = lambda coords: self.velFunction(coords, self.nu, t=time)
        fvort = lambda coords: self.vortFunction(coords, self.nu, t=time)
        self.vel = self.dom.applyFunctionVecToVec(self.bcNodes, fvel, self.vel, self.dim)
        self.vort = self.dom.applyFunctionVecToVec(self.bcNodes, fvort, self.vort, self.dim_w)

    @staticmethod
    def taylorGreenVel_2D(coord, nu, t=None):
        Lx = Ly = 1
        x_, y_ = 2 * pi * coord[0] / Lx, 2 * pi * coord[1] / Ly
        decay = exp(-4 * (pi ** 2) * nu * t * (1.0 / Lx ** 2 + 1.0 / Ly ** 2))
        return [cos(x_) * sin(y_) * decay, -sin(x_) * cos(y_) * decay]

    @staticmethod
    def taylorGreenVort_2D(coord, nu, t=None):
        Lx = Ly = 1
        x_, y_ = 2 * pi * coord[0] / Lx, 2 * pi * coord[1] / Ly
        decay = exp(-4 * (pi ** 2) * nu * t * (1.0 / Lx ** 2 + 1.0 / Ly ** 2))
        return [-2 * pi * (1.0 / Lx + 1.0 / Ly) * cos(x_) * cos(y_) * decay]
