"""The host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU).

``make -C graphsage_amd/csrc asan`` builds the library with its host code instrumented
(``tests/abi/libgsage_asan.so``) and ``tests/abi/abi_validate``, a plain C++ program that drives
every entry point's argument checks, hop / segment / draw table packing and size arithmetic with
arguments that must be refused and with empty inputs that must be accepted.  A sanitizer report
or a wrong return code fails the test.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_argument_checks_of_every_entry_point_are_clean_under_asan_and_ubsan():
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'graphsage_amd', 'csrc'), 'asan', '-j8'],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    env.pop('LD_PRELOAD', None)                                   # the sanitizer runtime must come first in the process
    res = subprocess.run([os.path.join(ROOT, 'tests', 'abi', 'abi_validate')], capture_output=True, text=True, env=env,
                         timeout=300)
    assert res.returncode == 0, (res.stdout[-3000:], res.stderr[-3000:])
    assert 'abi_validate ok' in res.stdout
    assert 'runtime error' not in res.stderr and 'AddressSanitizer' not in res.stderr


def test_every_declared_entry_point_is_exercised_by_the_validation_program():
    import re
    header = open(os.path.join(ROOT, 'include', 'gsage.h')).read()
    declared = set(re.findall(r'\b(gs_[a-z0-9_]+)\s*\(', header))
    used = set(re.findall(r'\b(gs_[a-z0-9_]+)\s*\(', open(os.path.join(ROOT, 'tests', 'abi', 'abi_validate.cpp')).read()))
    assert declared - used == set(), sorted(declared - used)
