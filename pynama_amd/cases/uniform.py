"""Uniform-flow case (mirrors src/cases/uniform.py:12-62): BC value [1,0(,0)] on the external
boundary, zero vorticity, exact solution v == const."""
from pynama_amd.cases.base_problem import FreeSlip


class UniformFlow(FreeSlip):
    def setUp(self):
        self.setUpGeneral()
        if self.dim == 2:
            self.cteValue = [1, 0]
        elif self.dim == 3:
            self.cteValue = [1, 0, 0]
        else:
            raise Exception("Wrong dim")
        # (the reference builds everything twice, uniform.py:14-30; once is enough)
        self.setUpBoundaryConditions()
        self.setUpEmptyMats()
        self.buildKLEMats()
        self.buildOperators()

    def computeInitialCondition(self, startTime):
        self.vort.set(0.0)

    def applyBoundaryConditions(self, time):
        self.vel.set(0.0)
        self.vel = self.dom.applyValuesToVec(self.bcNodes, self.cteValue, self.vel)

    def generateExactVecs(self, time=None):
        exactVel = self.mat.K.createVecRight()
        exactVort = self.mat.Rw.createVecRight()
        exactVel.setName(f"{self.caseName}-exact-vel")
        exactVort.setName(f"{self.caseName}-exact-vort")
        allNodes = self.dom.getAllNodes()
        exactVel = self.dom.applyValuesToVec(allNodes, self.cteValue, exactVel)
        exactVort.set(0.0)
        return exactVel, exactVort
