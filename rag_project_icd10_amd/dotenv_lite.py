"""`.env` loading for the stand-alone entry points. The reference calls python-dotenv's `load_dotenv()` at import time
(main.py:11, services/embedding_service.py:10, tools/build_database.py:11), so `python tools/build_database.py` picks
`EMBEDDING_MODEL_NAME`, `MILVUS_DB_PATH`, ... up from the project's `.env` (env.example). python-dotenv is not
installed in this deployment; this is the subset of its behaviour the reference relies on:

  * the file is the nearest `.env` walking UP from the directory of the calling module (python-dotenv's
    `find_dotenv(usecwd=False)`: `python /path/tools/build_database.py` finds the project's `.env` from any cwd) - here
    from this package's directory, so the project that contains the package; only when there is none, the nearest one
    walking up from the current directory (this deployment's addition);
  * `KEY=VALUE` lines, optional `export ` prefix, blank lines and `#` comments ignored, an unquoted value ends at an
    inline ` #`, matching single or double quotes are stripped (double-quoted values honour \\n and \\" escapes);
  * variables already present in the process environment are NOT overridden (load_dotenv's default).
"""
from __future__ import annotations

import os
from typing import Dict, Optional


def _walk_up(d: str, filename: str) -> str:
    d = os.path.abspath(d)
    while True:
        p = os.path.join(d, filename)
        if os.path.isfile(p):
            return p
        parent = os.path.dirname(d)
        if parent == d:
            return ""
        d = parent


def find_dotenv(start: Optional[str] = None, filename: str = ".env") -> str:
    if start:
        return _walk_up(start, filename)
    return _walk_up(os.path.dirname(os.path.abspath(__file__)), filename) or _walk_up(os.getcwd(), filename)


def parse_dotenv(text: str) -> Dict[str, str]:
    out: Dict[str, str] = {}
    for raw in text.splitlines():
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        if line.startswith("export "):
            line = line[len("export "):].lstrip()
        if "=" not in line:
            continue
        key, _, val = line.partition("=")
        key, val = key.strip(), val.strip()
        if not key:
            continue
        if len(val) >= 2 and val[0] == val[-1] and val[0] in "\"'":
            quote, val = val[0], val[1:-1]
            if quote == '"':
                val = val.replace("\\n", "\n").replace('\\"', '"')
        else:
            cut = val.find(" #")
            if cut >= 0:
                val = val[:cut].rstrip()
        out[key] = val
    return out


def load_dotenv(path: Optional[str] = None, override: bool = False) -> bool:
    """Returns True if a file was found and read."""
    path = path or find_dotenv()
    if not path or not os.path.isfile(path):
        return False
    with open(path, encoding="utf-8") as f:
        values = parse_dotenv(f.read())
    for k, v in values.items():
        if override or k not in os.environ:
            os.environ[k] = v
    return True
