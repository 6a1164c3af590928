"""Field output: /fields/<name> HDF5 datasets + the XDMF series of the reference's viewer
(src/viewer/paraviewer.py:18-58, src/viewer/xml_generator.py).  Host only: vectors are stand-ins."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from pynama_amd.common.comm import Comm
from pynama_amd.viewer import hdf5_writer
from pynama_amd.viewer.paraviewer import Paraviewer
from pynama_amd.viewer.xml_generator import XmlGenerator


class FakeVec:
    def __init__(self, name, arr):
        self.name, self.arr = name, np.asarray(arr, dtype=np.float64)

    def getName(self):
        return self.name

    def getArray(self):
        return self.arr

    def getSize(self):
        return self.arr.size


def test_hdf5_round_trip(tmp_path):
    p = str(tmp_path / "a.h5")
    a, b = np.linspace(0, 1, 11), np.random.default_rng(0).standard_normal(24)
    hdf5_writer.write_datasets(p, "fields", {"velocity": b, "vorticity": a})
    with open(p, "rb") as f:
        assert f.read(8) == b"\x89HDF\r\n\x1a\n"
    assert np.array_equal(hdf5_writer.read_dataset(p, "/fields/velocity"), b)
    assert np.array_equal(hdf5_writer.read_dataset(p, "/fields/vorticity"), a)
    with pytest.raises(RuntimeError):
        hdf5_writer.read_dataset(p, "/fields/nothing")


@pytest.mark.parametrize("dim", [2, 3])
def test_paraviewer_series(tmp_path, dim):
    n = 12
    rng = np.random.default_rng(dim)
    coords = FakeVec("NodeCoordinates", rng.random(n * dim))
    vel = FakeVec("velocity", rng.standard_normal(n * dim))
    vort = FakeVec("vorticity", rng.standard_normal(n * (1 if dim == 2 else 3)))
    d = str(tmp_path / "out")
    v = Paraviewer(dim, Comm(), d)
    v.saveMesh(coords)
    for step, t in ((0, 0.0), (7, 0.35)):
        v.saveData(step, t, vel, vort)
    v.writeXmf("case")
    assert np.array_equal(hdf5_writer.read_dataset(os.path.join(d, "mesh.h5"), "/fields/mesh"), coords.arr)
    assert np.array_equal(hdf5_writer.read_dataset(os.path.join(d, "vec-data-00007.h5"), "/fields/velocity"), vel.arr)
    assert np.array_equal(hdf5_writer.read_dataset(os.path.join(d, "vec-data-00000.h5"), "/fields/vorticity"), vort.arr)
    root = ET.parse(os.path.join(d, "case.xmf")).getroot()
    assert root.tag == "Xdmf" and root.get("Version") == "2.0"
    series = root.find("Domain/Grid")
    assert series.get("CollectionType") == "Temporal"
    grids = series.findall("Grid")
    assert [g.find("Time").get("Value") for g in grids] == ["0.0", "0.35"]
    g = grids[1]
    assert g.find("Topology").get("Dimensions") == str(n)
    assert g.find("Geometry").get("GeometryType") == ("XY" if dim == 2 else "XYZ")
    assert g.find("Geometry/DataItem").text == "mesh.h5:/fields/mesh"
    attrs = {a.get("Name"): a for a in g.findall("Attribute")}
    assert attrs["velocity"].get("AttributeType") == "Vector"
    join = attrs["velocity"].find("DataItem")
    assert join.get("Function") == XmlGenerator.getJoinString(dim) and join.get("Dimensions") == f"{n} {dim}"
    slabs = join.findall("DataItem")
    assert len(slabs) == dim
    assert slabs[1].findall("DataItem")[0].text == f"1 {dim} {n}"              # start, stride, count
    assert slabs[1].findall("DataItem")[1].text == "vec-data-00007.h5:/fields/velocity"
    # 2-D vorticity is a scalar field, 3-D a vector field (dimension test of saveStepInXML, paraviewer.py:52-57)
    assert attrs["vorticity"].get("AttributeType") == ("Scalar" if dim == 2 else "Vector")


def test_join_and_step_format():
    assert XmlGenerator.getJoinString(3) == "JOIN($0, $1, $2)" and XmlGenerator.getJoinString(2) == "JOIN($0, $1)"
    assert XmlGenerator.formatStep(42) == "00042"
    with pytest.raises(NotImplementedError):
        Paraviewer(2, Comm(1, 2), "/tmp/pynama_never").saveVec([], 0)
