"""Field output for ParaView (mirrors ``src/viewer/paraviewer.py``): ``mesh.h5`` with the node coordinates,
one ``vec-data-<step>.h5`` per saved step with ``/fields/<name>`` datasets, and an XDMF time series.

The reference goes through PETSc's HDF5 viewer; here the device vectors are copied to the host and written
with the image's libhdf5 (viewer/hdf5_writer.py).  Datasets are 1-D and interleaved (node-major, component-
minor) exactly as PETSc writes an unblocked Vec, which is what the XDMF hyperslabs expect.
One rank only: a collective write needs parallel HDF5, which this build does not bind."""
import os

import yaml

from pynama_amd.viewer import hdf5_writer
from pynama_amd.viewer.xml_generator import XmlGenerator


class Paraviewer:
    def __init__(self, dim, comm, saveDir=None):
        self.comm = comm
        self.saveDir = '.' if not saveDir else saveDir
        os.makedirs(self.saveDir, exist_ok=True)
        self.h5name = "vec-data"
        self.xmlWriter = XmlGenerator(dim, self.h5name)

    def _single_rank(self):
        if getattr(self.comm, "size", 1) != 1:
            raise NotImplementedError("field output runs on one rank (no parallel HDF5 in this build)")

    def saveMesh(self, coords, name='mesh'):
        self._single_rank()
        totalNodes = int(coords.getSize() / self.xmlWriter.dim)
        self.xmlWriter.setUpDomainNodes(totalNodes=totalNodes)
        self.xmlWriter.generateXMLTemplate()
        hdf5_writer.write_datasets(os.path.join(self.saveDir, "mesh.h5"), "fields", {name: coords.getArray()})

    def saveData(self, step, time, *vecs):
        self.saveVec(vecs, step)
        self.saveStepInXML(step, time, vecs=vecs)

    def saveVec(self, vecs, step):
        self._single_rank()
        hdf5_writer.write_datasets(os.path.join(self.saveDir, f"{self.h5name}-{step:05d}.h5"), "fields",
                                   {v.getName(): v.getArray() for v in vecs})

    def saveStepInXML(self, step, time, vec=None, vecs=None):
        dataGrid = self.xmlWriter.generateMeshData("mesh1")
        self.xmlWriter.setTimeStamp(time, dataGrid)
        for v in ([vec] if vec is not None else list(vecs)):
            if v.getSize() == self.xmlWriter.dimensions:
                self.xmlWriter.setScalarAttribute(v.getName(), step, dataGrid)
            else:
                self.xmlWriter.setVectorAttribute(v.getName(), step, dataGrid)

    def writeVTK(self, name, dm, step=None):
        raise NotImplementedError("VTK output of the DM (paraviewer.py:60-67) is PETSc-specific; use the XDMF series")

    def writeXmf(self, name):
        self.xmlWriter.writeFile(os.path.join(self.saveDir, name))

    def writeYaml(self, name, data):
        data['dir'] = self.saveDir
        with open(self.saveDir + '.yaml', 'w') as outfile:
            yaml.dump(data, outfile, default_flow_style=False)
