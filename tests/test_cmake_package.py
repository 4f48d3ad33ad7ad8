"""`find_package(aws-c-compression)` (reference CMakeLists.txt:58-94, cmake/aws-c-compression-config.cmake): the library
installs a package file under the same name with the same target, AWS::aws-c-compression; a consumer project written
for the reference (tests/cmake_consumer) configures, builds, links and runs against it unchanged."""
import os
import shutil
import subprocess

import pytest

import harness


@pytest.mark.skipif(shutil.which("cmake") is None, reason="no cmake in this image")
def test_consumer_finds_links_and_runs(tmp_path):
    if not os.path.exists(harness.PRODUCT_SO):
        import __graft_entry__

        __graft_entry__.build()
    prefix = tmp_path / "prefix"
    subprocess.check_call(["make", "-s", "-C", os.path.join(harness.REPO, "aws-c-compression_amd"), "install",
                           "PREFIX=%s" % prefix], stdout=subprocess.DEVNULL)
    assert (prefix / "lib" / "cmake" / "aws-c-compression" / "aws-c-compression-config.cmake").exists()
    build = tmp_path / "build"
    src = os.path.join(harness.REPO, "tests", "cmake_consumer")
    subprocess.check_call(["cmake", "-S", src, "-B", str(build), "-DCMAKE_PREFIX_PATH=%s" % prefix,
                           "-DCMAKE_BUILD_TYPE=Release"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", str(build)], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_LIBRARY_PATH="%s:%s" % (prefix / "lib", os.environ.get("LD_LIBRARY_PATH", "")))
    out = subprocess.run([str(build / "consumer")], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "linked and initialised" in out.stdout, out.stdout + out.stderr
