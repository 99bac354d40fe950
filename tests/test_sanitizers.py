"""CPU suite: AddressSanitizer + UndefinedBehaviorSanitizer builds of the host-side code, ThreadSanitizer over the eigenray scheduler (no GPU involved, GPU sanitizers are not
available on this pool): the set-up helpers of libgeoac_hip (geoac_host.cpp: .met and grid readers, spline slopes, fan enumeration,
grid table) and the plain-C oracle, each run over the fixtures and their error paths by a small driver (tests/sanitize/)."""
import os
import subprocess

import pytest

import harness as H
import rngdep_data as RD

SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def test_host_helpers_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "san_host")
    subprocess.check_call(["g++", "-std=c++17"] + SAN + ["-o", exe, os.path.join(H.ROOT, "tests", "sanitize", "san_host_driver.cpp"),
                                                         os.path.join(H.ROOT, "geoac_amd", "csrc", "geoac_host.cpp")])
    grid = RD.write_grid(str(tmp_path / "g"), short_paths=False)
    r = subprocess.run([exe, H.TOYATMO, *grid], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and b"san_host_driver ok" in r.stdout, r.stdout.decode()[-3000:]


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "san_oracle")
    subprocess.check_call(["gcc", "-std=c99"] + SAN + ["-ffp-contract=off", "-o", exe, os.path.join(H.ROOT, "tests", "sanitize", "san_oracle_driver.c"),
                                                       os.path.join(H.ROOT, "oracle", "geoac_oracle.c"), "-lm"])
    r = subprocess.run([exe, H.TOYATMO], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and b"san_oracle_driver ok" in r.stdout, r.stdout.decode()[-3000:]


def test_eigenray_scheduler_under_tsan(tmp_path):
    """the round scheduler of geoac_eigenray.cpp and its per-group worker threads (one context clone each) under ThreadSanitizer, on a
    closed-form stub of the fan ABI (tests/sanitize/tsan_eig_driver.cpp): 12 receivers x 2 bounce counts, scans and refinements sharing rounds"""
    exe = str(tmp_path / "tsan_eig")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-o", exe,
                           os.path.join(H.ROOT, "tests", "sanitize", "tsan_eig_driver.cpp"), os.path.join(H.ROOT, "geoac_amd", "csrc", "geoac_eigenray.cpp")])
    r = subprocess.run([exe], env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "tsan_eig_driver ok" in out and "ThreadSanitizer" not in out, out[-3000:]
