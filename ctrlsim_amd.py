"""Import alias: `import ctrlsim_amd` loads the package that lives in `ctrl-sim_amd/`
(a hyphen is not a legal Python identifier, the directory name is fixed by the repo layout)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "ctrl-sim_amd")]
__package__ = __name__
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
