#!/usr/bin/env python3
"""What distinguishes a box where the overlapped forms of the 8 192-env rollout gain from one where they lose (r06_experiments item 5)?
Prints the box's static description (rocm-smi: firmware, partition modes, power cap, clocks, memory vendor), then times the fused
launch in stream order against rollout chunks (8 192 envs + K = 4 ring, alternating, each in a loop of its own) while a thread samples
the hwmon / dpm files of the GPU (shader clock, memory clock, socket power, temperatures) -- so that the two states can be laid beside
the clocks they ran at.   usage: box_probe.py [envs] [steps per loop] [rounds]"""
import glob
import hashlib
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=60).stdout
    except Exception as ex:   # noqa: BLE001
        return "(%s)" % ex


def static():
    out = sh("rocm-smi --showhw --showfwinfo --showcomputepartition --showmemorypartition --showmaxpower --showperflevel --showmemvendor --showclocks --showvoltage --showpower --showtemp 2>&1")
    keep = []
    for ln in out.splitlines():
        if "Unique" in ln or "Serial" in ln:
            continue
        if ln.strip() and not ln.startswith("="):
            keep.append(ln.rstrip())
    uid = sh("rocm-smi --showuniqueid 2>&1")
    print("box id (sha of the GPU's unique id): %s" % hashlib.sha256(uid.encode()).hexdigest()[:12])
    print("\n".join(keep))
    print("uname: " + sh("uname -r").strip() + "   driver: " + sh("cat /sys/module/amdgpu/version 2>/dev/null").strip())


def hwmon_files():
    files = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            if open(card + "/vendor").read().strip() != "0x1002":
                continue
        except OSError:
            continue
        for hw in glob.glob(card + "/hwmon/hwmon*"):
            for f in sorted(glob.glob(hw + "/freq*_input") + glob.glob(hw + "/power*_average") + glob.glob(hw + "/power*_input") + glob.glob(hw + "/temp*_input")):
                lab = f.replace("_input", "_label").replace("_average", "_label")
                try:
                    name = open(lab).read().strip()
                except OSError:
                    name = os.path.basename(f)
                files[os.path.basename(card[:-7]) + ":" + name + ":" + os.path.basename(f)] = f
        break                                   # the one GPU of the box
    return files


class Sampler(threading.Thread):
    def __init__(self, files):
        super().__init__(daemon=True)
        self.files, self.rows, self.stop = files, [], False

    def run(self):
        while not self.stop:
            row = {}
            for k, f in self.files.items():
                try:
                    row[k] = int(open(f).read().strip())
                except (OSError, ValueError):
                    pass
            self.rows.append(row)
            time.sleep(0.05)

    def summary(self):
        out = []
        for k in self.files:
            v = sorted(r[k] for r in self.rows if k in r)
            if v:
                out.append("%s min %d med %d max %d" % (k.split(":", 1)[1], v[0], v[len(v) // 2], v[-1]))
        return "%d samples; " % len(self.rows) + "; ".join(out)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    static()
    files = hwmon_files()
    print("sampled files: " + ", ".join(sorted(files)))
    from toybox_amd import Engine, _abi, hip
    shift = int(os.environ.get("BP_SHIFT_KB", "0"))       # an allocation in front of the engine's: every buffer of it lands elsewhere
    if shift:
        hip.malloc(shift << 10)
    e = Engine(os.environ.get("BP_GAME", "breakout"), n)
    print("device: %s" % (e.device_identity(),))
    e.seed(1234); e.new_game()
    G = int(os.environ.get("BP_GATHER", "4"))                # K of the record ring (0: no gather)
    if G:
        e.set_option(_abi.OPT_GATHER_EVERY, G)
        e.gather_init(1, 0, e.gather_unique_id())
    e.set_option(_abi.OPT_FUSED_OVERLAP, _abi.FUSED_OVERLAP_OFF)
    st = hip.Stream()
    t = 0
    import ctypes as C
    clk = None
    so = os.path.join(ROOT, "scripts", "ubench", "libclock_probe.so")
    if os.path.exists(so) and os.environ.get("BP_CLOCK", "1") == "1":
        clk = C.CDLL(so)
        clk.clkp_start.argtypes = [C.c_int, C.c_int]
        clk.clkp_stop.argtypes = [C.POINTER(C.c_double), C.c_int]
    mhz = (C.c_double * 4096)()

    CK = G if G else int(os.environ.get("BP_K", "4"))        # frames per chunk (with a ring: its K)
    issued = [0.0]

    def loop(form, k):
        nonlocal t
        for _ in range(0, k, CK):
            if form != "order":
                e.rollout_synthetic(1337, t, CK, channels=3, auto_reset=True, stream=st.ptr)
            else:
                for j in range(CK):
                    e.render_step_synthetic(1337, t + j, channels=3, auto_reset=True, stream=st.ptr)
                    if G:
                        e.gather(stream=st.ptr)
            t += CK
        issued[0] = time.perf_counter()                 # (every call is queued: the host's share of the loop ends here)
        if form != "order":
            e.device_buffer(_abi.BUF_ROLLOUT_FRAMES)     # (the lazy join: the caller's stream behind the chunk's lanes)
        if G:
            e.gather_wait(stream=st.ptr)
        st.synchronize()                               # (not a device-wide wait: the clock probe's kernel is still running)

    # the forms: "order" = the fused launch in stream order; a number = tbx_rollout_synthetic with TBX_OPT_ROLLOUT_CHUNKS set to it
    # (1 on, 3 a rasteriser launch per frame on two lanes, 4 one per chunk on one lane)
    forms = os.environ.get("BP_FORMS", "order,1").split(",")
    for form in forms:
        e.set_option(_abi.OPT_ROLLOUT_CHUNKS, _abi.ROLLOUT_CHUNKS_OFF if form == "order" else int(form))
        loop(form, 400)
    for r in range(rounds):
        for form in forms:
            e.set_option(_abi.OPT_ROLLOUT_CHUNKS, _abi.ROLLOUT_CHUNKS_OFF if form == "order" else int(form))
            s = Sampler(files)
            s.start()
            if clk:
                assert clk.clkp_start(20000, 20000) == 0
            t0 = time.perf_counter()
            loop(form, steps)
            dt = time.perf_counter() - t0
            s.stop = True
            s.join()
            note = ""
            if clk:
                k = clk.clkp_stop(mhz, 4096)
                v = sorted(mhz[i] for i in range(max(k, 0)))
                note = "shader clock over %d intervals of 20 ms: min %.0f med %.0f max %.0f MHz;  " % (k, v[0], v[len(v) // 2], v[-1]) if v else "(no clock samples: %d)  " % k
            print("round %d  %-6s  %.4f ms/step   (the host had queued everything after %.4f ms/step)   %s%s"
                  % (r, form, dt / steps * 1e3, (issued[0] - t0) / steps * 1e3, note, s.summary() if os.environ.get("BP_HWMON") else ""), flush=True)
    if os.environ.get("BP_ADDR"):
        print("addresses: " + ", ".join("%s %#x" % (k, e.device_buffer(v)[0]) for k, v in (("frames", _abi.BUF_ROLLOUT_FRAMES), ("packed", _abi.BUF_ROLLOUT_PACKED))))
    e.sync()
    e.close()


main()
