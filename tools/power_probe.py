#!/usr/bin/env python3
"""Samples the card's clock and power (rocm-smi, read-only) while a command runs: is the step power-limited?
usage: power_probe.py <out.txt> -- <command ...>"""
import subprocess
import sys
import threading
import time


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showuse", "--csv"], capture_output=True,
                             text=True, timeout=10).stdout
    except Exception as e:  # noqa: BLE001
        return "error: %r" % (e,)
    return out.strip().replace("\n", " | ")


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    stop = threading.Event()
    lines = []

    def loop():
        t0 = time.time()
        while not stop.is_set():
            lines.append("%.2f %s" % (time.time() - t0, sample()))
            stop.wait(0.25)

    th = threading.Thread(target=loop)
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join()
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    sys.exit(rc)


if __name__ == "__main__":
    main()
