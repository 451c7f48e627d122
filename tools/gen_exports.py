#!/usr/bin/env python3
"""Linker version script of libpasero_hip.so: exactly the functions include/pasero_hip.h declares are exported; the
launchers the translation units call across files (pk_gemm8p_launch, pk_set_error, ...) stay local to the library.
  python3 tools/gen_exports.py include/pasero_hip.h > exports.map"""
import re
import sys

names = sorted(set(re.findall(r'\b(pk_[a-z0-9_]+)\s*\(', open(sys.argv[1]).read())))
print('{\n  global:\n' + ''.join(f'    {n};\n' for n in names) + '  local:\n    *;\n};')
