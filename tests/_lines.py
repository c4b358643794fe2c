"""Every (name, error, tolerance) line a GPU check produced, appended to gpurun_out/gpu_check_lines.txt (gpurun_out/ is merged back from the GPU
box; the copies that matter are committed under profiles/): a green suite still shows HOW green."""
import os


def record(res, tag=""):
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "gpu_check_lines.txt"), "a") as fh:
            fh.write("".join(f"{'ok  ' if e <= t else 'FAIL'} {e:.3e} <= {t:.3e}  {tag}{n}\n" for n, e, t in res))
    except OSError:
        pass
