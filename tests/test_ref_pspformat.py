"""The phase-space file format of exp_amd/reader.py and oracle/psp_oracle.c against the reference's SOURCE TEXT.

The reference's readers cannot be compiled here (exputil/ParticleReader.cc needs yaml-cpp and HighFive) and its
output classes neither (src/*.cc need Eigen and yaml-cpp); the record code that CAN be is pinned byte for byte in
tests/test_ref_particle.py.  What this file adds is the part only the source text can give: the ORDER
and WIDTH of the stream writes of `Particle::writeBinary`, `ComponentHeader::write`, `Component::write_binary` and
`Component::write_binary_header`, the stream reads of `PParticle::read`, the members of `MasterHeader`, the magic
constants and the reader names -- each extracted from the function body where it lies and compared with the record layout
this package reads and writes.  A field added, dropped, reordered or widened in the reference fails here.

Runs only where /root/reference exists; nothing is copied.  CPU only."""
import os
import re

import numpy as np
import pytest

from exp_amd import reader as R

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference sources")


def body(relpath, signature_regex):
    """the text of the first function whose definition line matches, from its first `{` to the matching `}`"""
    lines = open(os.path.join(REF, relpath), errors="replace").read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.search(signature_regex, l) and not l.rstrip().endswith(";")]
    assert starts, (relpath, signature_regex)
    s = starts[0]
    depth, e, seen = 0, s, False
    while e < len(lines):
        code = re.sub(r'"(\\.|[^"\\])*"', '""', lines[e]).split("//")[0]
        depth += code.count("{") - code.count("}")
        seen = seen or "{" in code
        if seen and depth <= 0:
            break
        e += 1
    return "\n".join(lines[s:e + 1])


def writes(text):
    """[(expression written, sizeof argument)] of every `out->write((const char *)&x, sizeof(T))` in order"""
    return [(a.strip(" &()"), b.strip()) for a, b in
            re.findall(r"->write\(\s*\(const char\s*\*\)\s*([^,]+),\s*(?:\w+\s*\*\s*)?sizeof\(([^)]+)\)", text)]


def test_particle_record_order_and_widths():
    b = body("exputil/Particle.cc", r"^void Particle::writeBinary\(")
    w = writes(b)
    # index (unsigned long) first, only when indexing; then mass, pos, vel as float or double; pot + potext; attributes
    assert w[0] == ("indx", "unsigned long") and re.search(r"if \(indexing\)[^\n]*\n\s*out->write\(\(const char \*\)&\(indx\)", b)
    seq = [x for x, _ in w[1:]]
    assert seq == ["tf", "mass", "tf", "pv", "tf", "pv", "tf", "pot0", "it", "tf", "jt"]
    assert [t for _, t in w[1:]] == ["float", "double"] * 4 + ["int", "float", "double"]
    assert re.search(r"double pot0 = pot \+ potext;", b)
    order = [m.start() for m in (re.search(p, b) for p in (r"static_cast<float>\(mass\)", r"double pv = pos\[i\]",
                                                             r"double pv = vel\[i\]", r"pot0 = pot \+ potext",
                                                             r"for \(auto it : iattrib\)", r"for \(auto jt: dattrib\)"))]
    assert order == sorted(order)
    # the reader's side: PParticle::read takes them in the same order, 8 reals
    rd = body("include/ParticleReader.H", r"void read\(std::istream& in, int pcount")
    reads = re.findall(r"in\.read\(\(char \*\)&(\w+)(?:\[i\])?,\s*sizeof\((\w[\w ]*)\)\)", rd)
    assert reads == [("indx", "unsigned long"), ("_mass", "real"), ("_pos", "real"), ("_vel", "real"), ("_phi", "real"),
                     ("iattrib", "int"), ("_datr", "real")]
    assert "indx = pcount;" in rd
    sk = body("include/ParticleReader.H", r"void skip\(std::istream& in, int pcount")
    assert re.search(r"8\*sizeof\(real\)\s*\+\s*spos->comp\.niatr\*sizeof\(int\)\s*\+\s*spos->comp\.ndatr\*sizeof\(real\)", sk)
    # and the layout here
    for rs, real in ((4, "<f4"), (8, "<f8")):
        dt = R.psp_record_dtype(rs, True, 2, 3)
        assert dt.names == ("indx", "mass", "pos", "vel", "pot", "iattrib", "dattrib")
        assert dt.itemsize == 8 + 8 * rs + 2 * 4 + 3 * rs
        assert dt["indx"] == np.dtype("<u8") and dt["mass"] == np.dtype(real) and dt["iattrib"].base == np.dtype("<i4")
        assert R.psp_record_dtype(rs, False, 0, 0).itemsize == 8 * rs


def test_headers_and_magic():
    h = open(os.path.join(REF, "include/header.H")).read()
    m = re.search(r"class MasterHeader \{\s*public:(.*?)friend", h, re.S)
    assert re.findall(r"^\s*(double|int)\s+(\w+);", m.group(1), re.M) == [("double", "time"), ("int", "ntot"), ("int", "ncomp")]
    cw = body("exputil/header.cc", r"^bool ComponentHeader::write\(ostream")
    assert [x for x, _ in writes(cw)] == ["nbod", "niatr", "ndatr", "ninfochar", "info.get"]
    assert "ninfochar*sizeof(char)" in cw
    assert re.search(r"int ComponentHeader::defaultInfoSize = (\d+);", open(os.path.join(REF, "exputil/header.cc")).read()).group(1) \
        == str(R.DEFAULT_INFO_SIZE)
    pr = open(os.path.join(REF, "include/ParticleReader.H")).read()
    assert int(re.search(r"unsigned long magic = (0x[0-9a-f]+);", pr).group(1), 16) == R.PSP_MAGIC
    assert int(re.search(r"unsigned long mmask = (0x[0-9a-f]+);", pr).group(1), 16) == R.PSP_MMASK
    # the writer: magic + rsize as an unsigned long, then the header; the split master adds the int number of files and
    # the 1024-byte names
    wb = body("src/Component.cc", r"^void Component::write_binary\(ostream\* out, bool real4\)")
    assert re.search(r"unsigned long cmagic = magic \+ rsize;", wb)
    assert re.search(r"out->write\(\(const char\*\)&cmagic, sizeof\(unsigned long\)\);\s*if \(!header\.write\(out\)\)", wb)
    assert "outs << conf << std::endl" in wb and "p[k]->writeBinary(rsize, indexing, out)" in wb
    wh = body("src/Component.cc", r"^void Component::write_binary_header\(")
    assert re.search(r"&cmagic,\s*sizeof\(unsigned long\)\);\s*out->write\(\(const char\*\)&nfiles,\s*sizeof\(int\)\);\s*if \(!header\.write\(out\)\)", wh)
    assert re.search(r"const size_t PBUF_SIZ = (\d+);", wh).group(1) == str(R.SPL_NAME_SIZE)
    assert 'sout << prefix << "-" << n' in wh
    wp = body("src/Component.cc", r"^void Component::write_binary_particles\(std::ostream\* out, bool real4\)")
    assert re.search(r"unsigned int N = particles\.size\(\);\s*out->write\(\(const char\*\)&N, sizeof\(unsigned int\)\);", wp)
    ps = body("src/OutPSN.cc", r"^void OutPSN::Run\(")
    assert re.search(r"header\.time\s*=\s*tnow;\s*header\.ntot\s*=\s*comp->ntot;\s*header\.ncomp\s*=\s*comp->ncomp;", ps)
    assert "out.write((char *)&header, sizeof(MasterHeader));" in ps
    pq = body("src/OutPSQ.cc", r"^void OutPSQ::Run\(")
    assert 'cname << fname.str() << "_" << count++;' in pq and 'cname << "-" << myid;' in pq


def test_reader_names_and_dispatch_order():
    src = open(os.path.join(REF, "exputil/ParticleReader.cc")).read()
    m = re.search(r"ParticleReader::readerTypes\s*\{([^}]*)\}", src)
    assert re.findall(r'"(\w+)"', m.group(1)) == R.ParticleReader.readerTypes
    cr = body("exputil/ParticleReader.cc", r"^\s*ParticleReader::createReader\(")
    assert re.findall(r'reader\.find\("(\w+)"\) == 0', cr) == ["PSPout", "PSPspl", "PSPhdf5", "GadgetNative", "GadgetHDF5",
                                                              "TipsyNative", "TipsyXDR", "Bonsai1", "Bonsai"]
    assert re.search(r'Ptypes\s*\{"Gas", "Halo", "Disk", "Bulge", "Stars", "Bndry"\}', src) and R.GADGET_TYPES == \
        ["Gas", "Halo", "Disk", "Bulge", "Stars", "Bndry"]
    # Tipsy structs: every member a Real = float, in this order
    t = open(os.path.join(REF, "include/tipsy.H")).read()
    assert "using Real = float;" in t
    for name, dt in (("gas_particle", R.TIPSY_GAS), ("dark_particle", R.TIPSY_DARK), ("star_particle", R.TIPSY_STAR)):
        blk = re.search(r"struct " + name + r"\s*\{(.*?)\}\s*;", t, re.S).group(1)
        blk = blk.split("int ID()")[0]
        assert tuple(re.findall(r"Real\s+(\w+)\s*(?:\[MAXDIM\])?\s*;", blk)) == dt.names
    hd = re.search(r"struct Header\s*\{(.*?)#ifdef", t, re.S).group(1)
    assert re.findall(r"(double|int)\s+(\w+)\s*;", hd) == [("double", "time"), ("int", "nbodies"), ("int", "ndim"),
                                                         ("int", "nsph"), ("int", "ndark"), ("int", "nstar")]


def test_histogram_statements():
    """the three statements the float sums hinge on, and the normalisations"""
    fg = open(os.path.join(REF, "expui/FieldGenerator.cc")).read()
    assert 'ret["xy"](indx1, indx2) += p->mass * fac["xy"];' in fg
    assert "if (indx>=0 and indx<nbins) ret[indx] += p->mass;" in fg
    assert "ret[i] /= 4.0*pi/3.0*del*del*del*(3*i*(i+1) + 1);" in fg and "ret[i] /= pi*del*del*(2*i + 1);" in fg
    assert "int indx = floor((log(sqrt(rad)) - lrmin)/del);" in fg and "ret[i] /= exp(3.0*(lrmin + del*i)) * rf;" in fg
    assert "Eigen::VectorXf ret = Eigen::VectorXf::Zero(nbins);" in fg and "Eigen::MatrixXf vc2 = Eigen::MatrixXf::Zero(nbins, 3);" in fg
