"""The eight matrices of the no-slip / free-slip split (mirrors ``src/matrices/mat_ns.py:17-121``):
K, Rw, Rd, Krhs for the final solve and Kfs, Rwfs, Rdfs, Krhsfs for the free-slip pre-solve, all on the
shared device graph; filled by ONE device pass (``pyn_assemble_kle_noslip``)."""
import numpy as np

from pynama_amd.matrices.mat_generator import DeviceMat, Mat


class MatNS(Mat):
    def createEmptyKLEMats(self, rStart, rEnd, d_nnz_ind, o_nnz_ind, ind_d, ind_o, indicesDIR, indicesNS):
        self.ctx, self.dom = ind_d.ctx, ind_d.dom
        self.globalIndicesDIR = set(indicesDIR)
        self.globalIndicesNS = set(indicesNS)
        d, dw = self.dim, self.dim_w
        mk = lambda br, bc, name, rhs=False: DeviceMat(self.ctx, br, bc, name, rhs=rhs)
        # Krhs / Krhsfs: compact imposed-column matrices (rows next to an imposed node; laid out by the assembly for its DOF classes)
        self.K, self.Rw, self.Rd, self.Krhs = mk(d, d, "K"), mk(d, dw, "Rw"), mk(d, 1, "Rd"), mk(d, d, "Krhs", True)
        self.Kfs, self.Rwfs, self.Rdfs, self.Krhsfs = mk(d, d, "Kfs"), mk(d, dw, "Rwfs"), mk(d, 1, "Rdfs"), mk(d, d, "Krhsfs", True)
        self.mats = [self.K, self.Rw, self.Rd, self.Krhs, self.Kfs, self.Rwfs, self.Rdfs, self.Krhsfs]

    def assembleAll(self):
        for m in (self.K, self.Rw, self.Rd, self.Krhs):
            m.assemble()

    def dofClassesLocal(self, nsFaces, dirFaces):
        """uint8 [nLocal, dim]: 0 free, 1 tangential DOF of a node on a no-slip wall, 2 imposed in both
        solves (wall-normal DOF of a no-slip node, every DOF of a Dirichlet-wall node); the per-cell
        set algebra of NoSlipFreeSlip.buildKLEMats (base_problem.py:343-379) in closed form."""
        dom = self.dom
        cls = np.zeros((dom.nLocal, self.dim), dtype=np.uint8)
        for name in nsFaces:
            axis, _ = dom._border_axis[name]
            on = dom._on_border_mask(name)
            for dd in range(self.dim):
                if dd != axis:
                    cls[on, dd] = np.maximum(cls[on, dd], 1)
            cls[on, axis] = 2
        for name in dirFaces:
            cls[dom._on_border_mask(name), :] = 2
        return cls

    def assembleKLE(self, elem, nsFaces, dirFaces, alpha_d=1e3, alpha_w=1e2):
        for t in elem.deviceTables():
            self.ctx.tables_set(*t)
        self.dofClasses = self.dofClassesLocal(nsFaces, dirFaces)
        self.ctx.bc_set(self.dim, self.dofClasses)
        self.ctx.assemble_kle_noslip(alpha_d, alpha_w, [self.K.id, self.Krhs.id, self.Rw.id, self.Rd.id,
                                                        self.Kfs.id, self.Krhsfs.id, self.Rwfs.id, self.Rdfs.id])
