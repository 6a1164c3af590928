"""Problem skeleton on the GPU path.

Mirrors ``src/cases/base_problem.py``: ``BaseProblem`` (config parsing :18-44,97-115, ``setUp``
:46-51, ``setUpDomain`` :53-78, ``setUpElement`` :80-83, ``setUpSolver`` :274-289) and ``FreeSlip``
(``setUpEmptyMats`` :463-477, ``buildKLEMats`` :499-552, ``solveKLE`` :479-481, ``getKLEError``
:483-497).  The per-cell Python loop of ``buildKLEMats`` is ONE fused device pass here.

Also here, from SURVEY.md section 8(f): the operator chain of ``evalRHS`` (f1: ``buildOperators`` / ``computeVtensV``,
:132-140, 212-252) and ``NoSlipFreeSlip`` (f2, :315-454); ``saveStep`` writes the HDF5 / XDMF pair of f4.
Not built (out of scope): the PETSc TS time integrator that would call ``evalRHS``.
"""
import logging

import numpy as np

from pynama_amd.common.comm import get_world
from pynama_amd.common.options import Options
from pynama_amd.common.timer import Timer
from pynama_amd.domain.dmplex import DMPlexDom
from pynama_amd.elements.spectral import Spectral
from pynama_amd.matrices.mat_generator import Mat, Operators
from pynama_amd.matrices.mat_ns import MatNS
from pynama_amd.solver.ksp_solver import KspSolver
from pynama_amd.vectors import Vec


class BaseProblem(object):
    def __init__(self, config, **kwargs):
        self.comm = get_world()
        self.timerTotal = Timer()
        self.timerTotal.tic()
        self.timer = Timer()
        if 'case' in kwargs:
            case = kwargs['case']
        else:
            case = Options().getString('case', 'uniform')
        self.config = config
        self.logger = logging.getLogger(f"[{self.comm.rank}] {self.config.get('name')}")
        self.case = case
        self.caseName = self.config.get("name")
        self.readDomainData(kwargs)
        self.readMaterialData()
        self.opts = kwargs
        if 'boundary-conditions' in self.config:
            self.readBoundaryCondition(self.config.get("boundary-conditions"))

    def setUp(self):
        self.setUpGeneral()
        self.setUpBoundaryConditions()
        self.setUpEmptyMats()
        self.buildKLEMats()
        self.buildOperators()

    def setUpDomain(self):
        domain = self.config.get("domain")
        self.dom = None
        if "box-mesh" in domain:
            self.logger.info("Creating dom with box Mesh")
            meshData = domain.get('box-mesh')
            self.meshType = "box-mesh"
            opts = {k: v for k, v in self.opts.items() if k in ('nelem', 'lower', 'upper', 'jitter', 'jitterSeed')}
            self.dom = DMPlexDom(boxMesh=meshData, **opts)
        elif "gmsh-file" in domain:
            self.meshType = 'gmsh'
            self.dom = DMPlexDom(fileName=domain.get('gmsh-file'), comm=self.comm)
        self.dim = self.dom.getDimension()
        self.dim_w = 1 if self.dim == 2 else 3
        self.dim_s = 3 if self.dim == 2 else 6
        self.ngl = self.opts['ngl'] if "ngl" in self.opts else domain['ngl']
        self.dom.setFemIndexing(self.ngl)
        if not self.comm.rank:
            self.logger.info("DMPlex dom created")

    def setUpElement(self):
        if getattr(self.dom, "cellType", "tensor") == "simplex":    # imported triangles / tetrahedra
            from pynama_amd.elements.simplex import Simplex
            self.elemType = Simplex(self.dim)
        else:
            self.elemType = Spectral(self.ngl, self.dim)
        if not self.comm.rank:
            self.logger.info(f"{self.dim}-D ngl:{self.ngl} Spectral element created")

    def setUpBoundaryConditions(self):
        self.dom.setLabelToBorders()
        self.dom.setBoundaryCondition()
        if not self.comm.rank:
            self.logger.info("Boundary Conditions setted up")

    def readMaterialData(self):
        materialData = self.config.get("material-properties")
        self.rho = materialData['rho']
        self.mu = materialData['mu']
        self.nu = self.mu / self.rho

    def readDomainData(self, kwargs):
        domain = self.config.get("domain")
        box = domain.get('box-mesh', {}) if domain else {}
        if "nelem" in kwargs:
            self.nelem = kwargs['nelem']
        elif "box-mesh" in domain:
            self.nelem = box['nelem']
        elif "gmsh-file" in domain:
            # the reference stops here ("No Gmsh Implemented", base_problem.py:106); dimension and ngl
            # come from the file in setUpDomain
            from pynama_amd.domain.gmsh import read_msh
            self.nelem = [0] * read_msh(domain['gmsh-file'])["dim"]
        else:
            raise Exception("No Gmsh Implemented")
        self.dim = len(self.nelem)
        self.lower = list(kwargs.get('lower', box.get('lower', [0] * self.dim)))[:self.dim]
        self.upper = list(kwargs.get('upper', box.get('upper', [1] * self.dim)))[:self.dim]
        self.dim_w = 1 if self.dim == 2 else 3
        self.dim_s = 3 if self.dim == 2 else 6
        self.ngl = kwargs['ngl'] if "ngl" in kwargs else domain['ngl']

    def createMesh(self, saveMesh=None):
        """base_problem.py:117-125.  Unlike the reference, setUp() does not write `mesh.h5` into the working
        directory unless asked to (`saveMesh=True`, or `save-output: true` in the yaml)."""
        self.dom.computeFullCoordinates(self.elemType)
        self.viewer = None
        if saveMesh is None:
            saveMesh = bool(self.config.get("save-output", False))
        if saveMesh:
            self.getViewer().saveMesh(self.dom.fullCoordVec)
            self._meshSaved = True
        if not self.comm.rank:
            self.logger.info("Mesh created")

    def getViewer(self):
        if getattr(self, "viewer", None) is None:
            from pynama_amd.viewer.paraviewer import Paraviewer
            self.viewer = Paraviewer(self.dim, self.comm, self.config.get("save-dir"))
            self._meshSaved = False
        return self.viewer

    def saveStep(self, step, time, *extra):
        """what the reference's converged-step callbacks write (base_problem.py:174-181, 201-202): the fields of
        this step as /fields/<name> datasets + the XDMF time series"""
        viewer = self.getViewer()
        if not self._meshSaved:
            viewer.saveMesh(self.dom.fullCoordVec)
            self._meshSaved = True
        viewer.saveData(step, time, self.vel, self.vort, *extra)
        viewer.writeXmf(self.caseName)

    def setUpGeneral(self):
        self.setUpDomain()
        self.setUpElement()
        self.createMesh()
        self.bcNodes = self.dom.getNodesFromLabel("External Boundary")

    def buildOperators(self):
        """SrT / DivSrT / Curl operators (base_problem.py:132-140) in three device passes."""
        self.operator.assembleOperators(self.elemType)
        if not self.comm.rank:
            self.logger.info("Operators Matrices builded")

    def evalRHS(self, ts, t, Vort, f):
        """Right-hand side of the vorticity transport equation (the role of base_problem.py:212-232):
        f = Curl( Div( 2 mu S(v) - rho v (x) v ) / rho ) with v from the KLE solve for `Vort`.  Four device products and
        two fused vector passes; the work vectors live as long as the problem."""
        self.solveKLE(t, Vort)
        ctx, stress = self.dom.ctx, self._Aux1
        self.computeVtensV()
        self.operator.SrT.mult(self.vel, stress)
        ctx.vec_axpby(stress.id, 2.0 * self.mu, stress.id, -self.rho, self._VtensV.id)      # 2 mu S(v) - rho v (x) v, one pass
        if getattr(self, "_divStress", None) is None:
            self._divStress = self.vel.duplicate()
        self.operator.DivSrT.mult(stress, self._divStress)
        self._divStress.scale(1.0 / self.rho)
        self.operator.Curl.mult(self._divStress, f)

    def solveKLE(self, time, vort):
        pass

    def buildKLEMats(self):
        pass

    def computeInitialCondition(self, startTime):
        pass

    def applyBoundaryConditions(self, time):
        pass

    def readBoundaryCondition(self, bc=None):
        pass

    def setUpSolver(self):
        self.solver = KspSolver()
        self.solver.createSolver(self.mat.K, self.comm)
        self.vel = self.mat.K.createVecRight()
        self.vel.setName("velocity")
        self.vort = self.mat.Rw.createVecRight()
        self.vort.setName("vorticity")
        self.vort.set(0.0)
        self._VtensV = Vec(self.dom.ctx, self.dim_s)
        self._Aux1 = Vec(self.dom.ctx, self.dim_s)

    def computeVtensV(self, vec=None):
        """v (x) v in Voigt-like order (base_problem.py:234-252), on the device."""
        src = vec if vec is not None else self.vel
        self.dom.ctx.vec_vtensv(src.id, self._VtensV.id)

    def setUpEmptyMats(self):
        self.mat = None
        self.operator = None

    def view(self):
        print(f"Case: {self.case}")
        print(f"Domain: {self.dom.view()} ")
        print(f"NGL: {self.dom.getNGL() }")


class NoSlipFreeSlip(BaseProblem):
    """Two-solve KLE with no-slip walls (base_problem.py:300-454): a free-slip pre-solve on K + Kfs whose
    wall vorticity feeds the final solve on K."""

    def setUpEmptyMats(self):
        self.mat = MatNS(self.dim, self.comm)
        self.operator = Operators(self.dim, self.comm)
        rStart, rEnd, d_nnz_ind, o_nnz_ind, ind_d, ind_o = self.dom.getMatIndices()
        self.globalNodesDIR = self.dom.getGlobalIndicesDirichlet()
        globalNodesNS = self.dom.getGlobalIndicesNoSlip()
        self.mat.createEmptyKLEMats(rStart, rEnd, d_nnz_ind, o_nnz_ind, ind_d, ind_o, self.globalNodesDIR, globalNodesNS)
        if not self.comm.rank:
            self.logger.info("Empty KLE Matrices created")
        self.operator.createAll(rStart, rEnd, d_nnz_ind, o_nnz_ind, graph=ind_d)

    def setUpSolver(self):
        super().setUpSolver()
        self.solverFS = KspSolver()
        self.solverFS.createSolver(self.mat.K + self.mat.Kfs, self.comm)       # base_problem.py:318
        self.velFS = self.vel.copy()

    def solveKLE(self, time, vort):                                            # base_problem.py:321-327
        self.applyBoundaryConditions()
        self.solverFS(self.mat.Rw * vort + self.mat.Rwfs * vort + self.mat.Krhsfs * self.vel, self.velFS)
        self.applyBoundaryConditionsFS()
        vort = self.operator.Curl * self.velFS
        self.solver(self.mat.Rw * vort + self.mat.Krhs * self.vel, self.vel)

    def buildKLEMats(self):
        """base_problem.py:329-454 as one device pass over all cells (the reference integrates cell 0 only
        and reuses it, :333-334; here every cell is integrated)."""
        self.mat.assembleKLE(self.elemType, self._nsFaces, self._dirFaces)
        self.mat.assembleAll()
        if not self.comm.rank:
            self.logger.info("KLE Matrices builded")


class FreeSlip(BaseProblem):
    def generateExactVecs(self, time):
        return 0, 0

    def setUpEmptyMats(self):
        self.mat = Mat(self.dim, self.comm)
        self.operator = Operators(self.dim, self.comm)
        rStart, rEnd, d_nnz_ind, o_nnz_ind, ind_d, ind_o = self.dom.getMatIndices()
        globalIndicesDIR = self.dom.getGlobalIndicesDirichlet()
        d_nnz_ind_op = d_nnz_ind.copy()
        self.mat.createEmptyKLEMats(rStart, rEnd, d_nnz_ind, o_nnz_ind, ind_d, ind_o, globalIndicesDIR)
        if not self.comm.rank:
            self.logger.info("Empty KLE Matrices created")
        self.operator.createAll(rStart, rEnd, d_nnz_ind_op, o_nnz_ind, graph=ind_d)

    def solveKLE(self, time, vort):
        self.applyBoundaryConditions(time)
        self.solver(self.mat.Rw * vort + self.mat.Krhs * self.vel, self.vel)     # base_problem.py:481

    def getKLEError(self, viscousTimes=None, startTime=0.0, endTime=1.0, steps=10):
        """l2 error of the KLE velocity against the case's exact fields at the times t = tau^2 / (4 nu) of the viscous times `tau`
        (what the reference's convergence charts plot, base_problem.py:483-497)."""
        taus = np.linspace(startTime, endTime, steps, endpoint=False) if viscousTimes is None else np.asarray(viscousTimes, dtype=float)
        errors = []
        for t in taus ** 2 / (4.0 * self.nu):
            exactVel, exactVort = self.generateExactVecs(t)
            self.solveKLE(t, exactVort)
            errors.append((exactVel - self.vel).norm(norm_type=2))
        return errors

    def buildKLEMats(self):
        """base_problem.py:499-552 as one fused device pass: quadrature of every cell
        (spectral.py:89-157), scatter with Dirichlet elimination, unit diagonal (:549)."""
        # like the reference (locK, locRw, _ = getElemKLEMatrices, :529), the free-slip problem does not fill Rd
        self.mat.assembleKLE(self.elemType, alpha_d=1e3, alpha_w=1e2, with_rd=False)
        self.mat.setIndices2One(self.mat.globalIndicesDIR)
        self.mat.assembleAll()
        if not self.comm.rank:
            self.logger.info("KLE Matrices builded")
