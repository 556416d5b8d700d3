"""amd-smi sampling shared by tools/clock_probe.py and tools/step_power_probe.py"""
import json
import subprocess


def sample():
    """(power W, gfx clock MHz) or None"""
    try:
        out = subprocess.run(["amd-smi", "metric", "-g", "0", "--power", "--clock", "--json"], capture_output=True, text=True, timeout=10).stdout
        j = json.loads(out)
        j = j[0] if isinstance(j, list) else j.get("gpu_data", [j])[0] if isinstance(j, dict) else j
        pw = j.get("power", {})
        p = pw.get("socket_power", pw.get("average_socket_power", None))
        p = p.get("value") if isinstance(p, dict) else p
        clk = j.get("clock", {})
        gfx = [v.get("clk", {}).get("value") if isinstance(v.get("clk"), dict) else v.get("clk") for k, v in clk.items() if k.startswith("gfx") and isinstance(v, dict)]
        gfx = [float(x) for x in gfx if isinstance(x, (int, float))]
        return (float(p) if p not in (None, "N/A") else None, sum(gfx) / len(gfx) if gfx else None, max(gfx) if gfx else None)
    except Exception as e:              # noqa: BLE001 - a diagnostic tool: say what happened and go on
        return ("ERR", repr(e)[:200], None)
