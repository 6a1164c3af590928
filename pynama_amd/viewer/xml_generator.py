"""XDMF description of the saved time series (mirrors ``src/viewer/xml_generator.py``): a temporal collection
of point clouds whose geometry is ``mesh.h5:/fields/mesh`` and whose attributes point into
``<h5name>-<step>.h5:/fields/<name>``; vector attributes are a JOIN of per-component hyperslabs of the
interleaved dataset (:62-97)."""
from xml.dom import minidom
from xml.etree.ElementTree import Element, SubElement, tostring


class XmlGenerator(object):
    def __init__(self, dim, h5name):
        self.root = Element('Xdmf', {'Version': '2.0'})
        self.dim = dim
        self.h5name = h5name
        self.dimensions = None

    def setUpDomainNodes(self, totalNodes=None, nodesPerDim=None):
        if totalNodes is not None:
            self.dimensions = int(totalNodes)
        else:
            assert len(nodesPerDim) == self.dim
            self.dimensions = 1
            for nodes in nodesPerDim:
                self.dimensions *= int(nodes)

    def generateXMLTemplate(self):
        self.domain = SubElement(self.root, 'Domain')
        self.grid = SubElement(self.domain, 'Grid', {'Name': 'TimeSeries', 'GridType': 'Collection',
                                                     'CollectionType': 'Temporal'})

    def generateMeshData(self, name):
        meshGrid = SubElement(self.grid, 'Grid', {'Name': name, 'GridType': 'uniform'})
        SubElement(meshGrid, 'Topology', {'TopologyType': 'Polyvertex', 'Dimensions': str(self.dimensions)})
        geometry = SubElement(meshGrid, 'Geometry', {'GeometryType': 'XY' if self.dim == 2 else 'XYZ'})
        data = SubElement(geometry, 'DataItem', {'Dimensions': str(self.dimensions * self.dim), 'NumberType': 'Float',
                                                 'Format': 'HDF'})
        data.text = "mesh.h5:/fields/mesh"
        return meshGrid

    def setTimeStamp(self, t, meshElem):
        SubElement(meshElem, 'Time', {'Value': str(t)})

    def _file(self, step, name):
        return f"{self.h5name}-{step:05d}.h5:/fields/{name}"

    def setScalarAttribute(self, name, step, meshGrid):
        attr = SubElement(meshGrid, 'Attribute', {'Name': name, 'AttributeType': 'Scalar', 'Center': 'Node'})
        data = SubElement(attr, 'DataItem', {'Dimensions': str(self.dimensions), 'NumberType': 'Float', 'Format': 'HDF'})
        data.text = self._file(step, name)

    def setVectorAttribute(self, name, step, meshGrid):
        attr = SubElement(meshGrid, 'Attribute', {'Name': name, 'AttributeType': 'Vector', 'Center': 'Node'})
        join = SubElement(attr, 'DataItem', {'ItemType': 'Function', 'Dimensions': f"{self.dimensions} {self.dim}",
                                             'Function': self.getJoinString(self.dim)})
        for dof in range(self.dim):
            self.setDataToAttribute(join, step, name, dof)

    def setDataToAttribute(self, attrData, step, name, dof):
        slab = SubElement(attrData, 'DataItem', {'ItemType': 'HyperSlab', 'Dimensions': str(self.dimensions),
                                                 'Name': f"{name}-{'XYZ'[dof]}"})
        sel = SubElement(slab, 'DataItem', {'Dimensions': '3 1', 'Format': 'XML'})
        sel.text = f"{dof} {self.dim} {self.dimensions}"              # start, stride, count
        data = SubElement(slab, 'DataItem', {'Dimensions': str(self.dimensions * self.dim), 'NumberType': 'Float',
                                             'Format': 'HDF'})
        data.text = self._file(step, name)

    def toString(self):
        return minidom.parseString(tostring(self.root, 'utf-8')).toprettyxml(indent=" ")

    def writeFile(self, nameFile):
        with open(f"{nameFile}.xmf", "w") as f:
            f.write(self.toString())

    def printify(self):
        print(self.toString())

    @staticmethod
    def formatStep(step):
        return f"{int(step):05d}"

    @staticmethod
    def getJoinString(dof):
        return "JOIN(" + ", ".join(f"${i}" for i in range(dof)) + ")"
