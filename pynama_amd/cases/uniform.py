"""Uniform-flow case: the velocity e_x imposed on the whole external boundary, zero vorticity, so the KLE solve must
return v = e_x everywhere.  API of the reference's case (src/cases/uniform.py: `UniformFlow` with `setUp`,
`computeInitialCondition`, `applyBoundaryConditions`, `generateExactVecs`); the matrices are built ONCE, by
`BaseProblem.setUp` (the reference assembles them twice in a row, uniform.py:14-30)."""
import numpy as np

from pynama_amd.cases.base_problem import FreeSlip


class UniformFlow(FreeSlip):
    def setUp(self):
        if self.dim not in (2, 3):
            raise Exception("Wrong dim")
        self.cteValue = np.eye(self.dim)[0].tolist()     # e_x: [1, 0] / [1, 0, 0]
        super().setUp()

    def _uniform(self, vec, nodes):
        vec.set(0.0)
        return self.dom.applyValuesToVec(nodes, self.cteValue, vec)

    def computeInitialCondition(self, startTime):
        self.vort.set(0.0)

    def applyBoundaryConditions(self, time):
        self.vel = self._uniform(self.vel, self.bcNodes)

    def generateExactVecs(self, time=None):
        vel, vort = self.mat.K.createVecRight(), self.mat.Rw.createVecRight()
        for vec, what in ((vel, "vel"), (vort, "vort")):
            vec.setName(f"{self.caseName}-exact-{what}")
        vort.set(0.0)
        return self._uniform(vel, self.dom.getAllNodes()), vort
