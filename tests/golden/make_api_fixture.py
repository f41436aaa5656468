#!/usr/bin/env python3
"""Names, parameter names and defaults of the reference's `libs.C_extension` API (libs/C_extension.pyi, the type stub of the pybind11
module cpp_src/tensor/bind.cpp:317-391) as DATA: tests/golden/c_extension_api.json.  Development container only (reads /root/reference)."""
import ast
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
tree = ast.parse(open("/root/reference/libs/C_extension.pyi").read())
api = {"functions": {}, "classes": {}, "attributes": []}
for node in tree.body:
    if isinstance(node, ast.FunctionDef):
        a = node.args
        names = [x.arg for x in a.args]
        defaults = [ast.literal_eval(d) for d in a.defaults]
        api["functions"][node.name] = {"params": names, "defaults": dict(zip(names[len(names) - len(defaults):], defaults))}
    elif isinstance(node, ast.ClassDef):
        api["classes"][node.name] = sorted(n.name for n in node.body if isinstance(n, ast.FunctionDef) and not n.name.startswith("_"))
    elif isinstance(node, ast.AnnAssign) and isinstance(node.target, ast.Name):
        api["attributes"].append(node.target.id)
json.dump(api, open(os.path.join(HERE, "c_extension_api.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(api, indent=1)[:1500])
