"""Node / DOF bookkeeping.  Mirrors ``src/domain/indices.py`` (``IndicesManager``): the
entity->node maps driven by PetscSection offsets (:66-114) are replaced by the closed-form
lattice numbering of ``domain/dmplex.py``; what remains is the DOF interleave (:90-92) and the
Dirichlet / no-slip node sets (:45-64)."""
import logging


class IndicesManager:
    def __init__(self, dim, ngl, comm):
        self.logger = logging.getLogger("[{}] IndicesManager Class".format(comm.rank))
        self.comm = comm
        self.dim = dim
        self._ngl = ngl
        self.__dirNodes = set()
        self.__nsNodes = set()

    def getNGL(self):
        return self._ngl

    def getNumCompAndNumDof(self, componentsPerField, numFields):
        numComp = [componentsPerField] * numFields
        nodesPerEntity = [1, self._ngl - 2, (self._ngl - 2) ** 2]
        if self.dim == 3:
            nodesPerEntity.append((self._ngl - 2) ** 3)
        numDof = [componentsPerField * nodes for nodes in nodesPerEntity]
        return numComp, numDof

    def setDirichletNodes(self, nodes: set):
        self.__dirNodes |= set(nodes)

    def getDirichletNodes(self):
        # every rank holds the GLOBAL set already (borders are closed-form): no allgather needed
        self.globalIndicesDIR = set(self.__dirNodes)
        return self.__dirNodes

    def setNoSlipNodes(self, nodes: set):
        self.__nsNodes |= set(nodes)

    def getNoSlipNodes(self):
        self.globalIndicesNS = set(self.__nsNodes)
        return self.__nsNodes

    def mapNodesToIndices(self, nodes, dof):
        """global DOF = node*dof + d (indices.py:90-92)"""
        return [x * dof + d for x in nodes for d in range(dof)]
