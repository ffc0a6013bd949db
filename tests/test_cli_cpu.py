"""CLI behaviour that is decided before any GPU work: clap-style argument groups
(src/cmd_extract.rs:33-62,102-110; src/cmd_tag.rs:29-66; tests of src/main.rs:61-293), the
log-flag conflict check (src/helpers.rs:172-200) and path helpers.  Runs without a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "merkurio_amd", "lib", "merkurio")


@pytest.fixture(scope="module", autouse=True)
def _built():
    from merkurio_amd import build
    build.build_all()


def rc(*args):
    p = subprocess.run([BIN, *args], capture_output=True)
    return p.returncode, p.stdout, p.stderr


def test_help_and_version():
    assert rc("--help")[0] == 0 and b"extract" in rc("--help")[1]
    assert rc("extract", "--help")[0] == 0 and rc("tag", "-h")[0] == 0
    assert rc("--version")[1].startswith(b"merkurio ")
    assert rc()[0] == 2 and rc("frobnicate")[0] == 2


def test_extract_argument_groups(golden):
    fa = os.path.join(golden, "fixtures/input/simple.fasta")
    assert rc("extract", "-s", "A")[0] == 2                                  # -i required
    assert rc("extract", "-i", fa)[0] == 2                                   # one of -s / -f required
    assert rc("extract", "-i", fa, "-s", "A", "-f", "k.txt")[0] == 2         # ... and only one
    assert rc("extract", "-i", fa, "-s", "A", "-q", "1", "-a")[0] == 2       # algorithm group
    assert rc("extract", "-i", fa, "-s", "A", "-I", "-L")[0] == 2            # case group
    assert rc("extract", "-i", fa, "-s", "A", "-L", "-U")[0] == 2
    assert rc("extract", "-i", fa, "-s", "A", "-c", "-r")[0] == 2            # preprocessing group
    assert rc("extract", "-i", fa, "-s", "A", "-S")[0] == 2                  # -S requires logging
    assert rc("extract", "-i", fa, "-s", "A", "-S", "-l", "-o", "x")[0] == 2  # -S conflicts with -o
    assert rc("extract", "-i", fa, "-s", "A", "--bogus")[0] == 2
    assert rc("extract", "-i", fa, "-s", "A", "-q", "x")[0] == 2
    assert rc("extract", "-i", fa, "-s")[0] == 2


def test_tag_argument_groups(golden):
    sam = os.path.join(golden, "fixtures/input/simple.sam")
    assert rc("tag", "-s", "A")[0] == 2
    assert rc("tag", "-i", sam, "-s", "A", "-m", "-v")[0] == 2               # matching group
    assert rc("tag", "-i", sam, "-s", "A", "-S")[0] == 2


def test_log_flag_conflicts(golden):
    fa = os.path.join(golden, "fixtures/input/simple.fasta")
    code, _, err = rc("extract", "-i", fa, "-s", "A", "-l", "-j")
    assert code == 1 and b"Cannot use both -l/--out-log and -j/--json-log" in err
    code, _, err = rc("extract", "-i", fa, "-s", "A", "-l")
    assert code == 1 and b"Cannot write log to stdout when normal output is also stdout" in err
    code, _, err = rc("extract", "-i", fa, "-f", os.path.join(golden, "data/kmers-empty.txt"), "-o", "x")
    assert code == 1 and b"No k-mers found" in err
    code, _, err = rc("extract", "-i", fa, "-f", "/nonexistent/kmers.txt", "-o", "x")
    assert code == 1 and b"File not found." in err
