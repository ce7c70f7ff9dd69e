import sys


def _early_preload():
    """Before click is even imported: if this is a plain clustering run, start loading the HIP library on a native thread
    (fastpath.run would do it a few tens of milliseconds later; a second start is a no-op)."""
    argv = []
    for a in sys.argv[1:]:  # "--opt=value" and "--opt value" are the same to click
        argv.extend(a.split("=", 1) if a.startswith("--") and "=" in a else [a])
    if "--input-file" not in argv or "--help" in argv:
        return
    try:
        i = argv.index("--max-dist") if "--max-dist" in argv else -1
        if i >= 0 and argv[i + 1].strip() == "0":
            return
        from . import _front

        _front.preload(argv[argv.index("--input-file") + 1])
    except Exception:  # never a reason to fail: the run proceeds without the head start
        pass


_early_preload()

from .console import main  # noqa: E402

if __name__ == "__main__":
    main()
