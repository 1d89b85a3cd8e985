"""Launch-plan / path overrides for the A/B tools. The library reads no environment variables: a variant is a dict with the
keys the round-1/2 tools used as environment names - FORGE_CONV_TILE (A..E), FORGE_CONV_KSPLIT (n), FORGE_WINOGRAD (0/1) - applied through
forge_amd.convops.force_plan / winograd, which hand the plan to the C-ABI calls as explicit arguments."""
import contextlib

from forge_amd import convops as co


@contextlib.contextmanager
def variant(env):
    tile = env.get("FORGE_CONV_TILE") or None
    ks = env.get("FORGE_CONV_KSPLIT")
    ks = int(ks) if ks not in (None, "") else None
    unknown = set(env) - {"FORGE_CONV_TILE", "FORGE_CONV_KSPLIT", "FORGE_WINOGRAD"}
    if unknown:
        raise KeyError("unknown variant switch(es) %s" % sorted(unknown))
    with contextlib.ExitStack() as st:
        if tile is not None or ks is not None:
            st.enter_context(co.force_plan(tile, ks))
        if "FORGE_WINOGRAD" in env:
            st.enter_context(co.winograd(str(env["FORGE_WINOGRAD"]) != "0"))
        yield


def apply_environ():
    """For the one-off scripts under tools/debug that still spell their variants as os.environ assignments: copy FORGE_CONV_TILE /
    FORGE_CONV_KSPLIT / FORGE_WINOGRAD from this PROCESS's environment into the convops overrides (the library never reads them)."""
    import os
    tile = os.environ.get("FORGE_CONV_TILE") or None
    ks = os.environ.get("FORGE_CONV_KSPLIT")
    co.STATE.plan_override = (tile, int(ks) if ks else None) if (tile or ks) else None
    co.STATE.winograd = os.environ.get("FORGE_WINOGRAD", "1") != "0"
