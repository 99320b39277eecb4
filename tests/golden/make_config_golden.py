"""Extracts every configuration default of the reference's plugin surface into tests/golden/ref_config.json by PARSING
(python `ast`, nothing is imported or executed) the reference sources:

  /root/reference/gaussctrl/gc_config.py:40-92     trainer flags, optimizer groups (lr / eps / scheduler), viewer
  /root/reference/gaussctrl/gc_pipeline.py:48-73   GaussCtrlPipelineConfig fields
  /root/reference/gaussctrl/gc_datamanager.py:54-66 GaussCtrlDataManagerConfig fields
  /root/reference/gaussctrl/gc_model.py:39-50      GaussCtrlModelConfig fields
  /root/reference/gaussctrl/gc_trainer.py:42-47    GaussCtrlTrainerConfig fields
  /root/reference/pyproject.toml:38-42             entry points

The JSON is data (names and literal values); tests/test_plugin_config.py diff-checks gaussctrl_amd's configs against it.
usage: python tests/golden/make_config_golden.py
"""
import ast
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def lit(node):
    """literal value of an AST expression (numbers, strings, dicts, simple arithmetic, shifts); None for anything else"""
    try:
        return ast.literal_eval(node)
    except Exception:  # noqa: BLE001
        pass
    if isinstance(node, ast.BinOp):
        a, b = lit(node.left), lit(node.right)
        if a is None or b is None:
            return None
        if isinstance(node.op, ast.Div):
            return a / b
        if isinstance(node.op, ast.Mult):
            return a * b
        if isinstance(node.op, ast.LShift):
            return a << b
    return None


def call_kwargs(call):
    out = {}
    for kw in call.keywords:
        v = kw.value
        if isinstance(v, ast.Call):
            out[kw.arg] = {"__call__": ast.unparse(v.func), **call_kwargs(v)}
        elif isinstance(v, ast.Dict):
            d = {}
            for k, x in zip(v.keys, v.values):
                key = lit(k)
                if isinstance(x, ast.Dict):
                    d[key] = {lit(kk): ({"__call__": ast.unparse(xx.func), **call_kwargs(xx)} if isinstance(xx, ast.Call) else lit(xx))
                              for kk, xx in zip(x.keys, x.values)}
                elif isinstance(x, ast.Call):
                    d[key] = {"__call__": ast.unparse(x.func), **call_kwargs(x)}
                else:
                    d[key] = lit(x)
            out[kw.arg] = d
        else:
            out[kw.arg] = lit(v)
    return out


def dataclass_fields(path, cls):
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            out = {}
            for st in node.body:
                if isinstance(st, ast.AnnAssign) and isinstance(st.target, ast.Name) and not st.target.id.startswith("_"):
                    v = lit(st.value) if st.value is not None else None
                    if v is None and st.value is not None and not (isinstance(st.value, ast.Constant) and st.value.value is None):
                        v = {"__expr__": ast.unparse(st.value)}
                    out[st.target.id] = v
            return {"bases": [ast.unparse(b) for b in node.bases], "fields": out}
    raise KeyError(cls)


def main():
    tree = ast.parse(open(f"{REF}/gaussctrl/gc_config.py").read())
    spec = None
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "gaussctrl_method":
            spec = node.value
    top = call_kwargs(spec)
    pp = open(f"{REF}/pyproject.toml").read()
    eps = dict(re.findall(r"^\s*([\w-]+)\s*=\s*['\"]([\w.:]+)['\"]\s*$", pp.split("[project.entry-points", 1)[1], flags=re.M))
    out = {
        "method_specification": top,
        "GaussCtrlPipelineConfig": dataclass_fields(f"{REF}/gaussctrl/gc_pipeline.py", "GaussCtrlPipelineConfig"),
        "GaussCtrlDataManagerConfig": dataclass_fields(f"{REF}/gaussctrl/gc_datamanager.py", "GaussCtrlDataManagerConfig"),
        "GaussCtrlModelConfig": dataclass_fields(f"{REF}/gaussctrl/gc_model.py", "GaussCtrlModelConfig"),
        "GaussCtrlTrainerConfig": dataclass_fields(f"{REF}/gaussctrl/gc_trainer.py", "GaussCtrlTrainerConfig"),
        "entry_points": eps,
    }
    with open(os.path.join(HERE, "ref_config.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
