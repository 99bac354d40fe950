"""Runs the reference's own CLI binaries (oracle/_ref/GeoAc*, compiled from /root/reference by oracle/Makefile)
on small fans and stores their output FILES (data) under tests/golden/cli/<set>/.  Run here only."""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.path.join(ROOT, "oracle", "_ref")

CASES = {
    "global": ("GeoAcGlobal", ["theta_min=15", "theta_max=35", "theta_step=20", "phi_min=-90", "phi_max=-45", "phi_step=45",
                               "bounces=1", "WriteCaustics=True", "WriteAtmo=True"]),
    "3d": ("GeoAc3D", ["theta_min=15", "theta_max=35", "theta_step=20", "phi_min=-90", "phi_max=-45", "phi_step=45",
                       "bounces=1", "WriteCaustics=True", "WriteAtmo=True"]),
    "2d": ("GeoAc2D", ["theta_min=5", "theta_max=35", "theta_step=15", "azimuth=-80", "bounces=1", "WriteCaustics=True"]),
    "global_norays": ("GeoAcGlobal", ["theta_min=5", "theta_max=45", "theta_step=8", "azimuth=37", "WriteRays=False", "CalcAmp=False",
                                      "lat_src=41.131", "lon_src=-112.896", "z_src=1.0", "freq=0.5", "rng_max=900"]),
    # BASELINE config 1, the exact command: GeoAc2D -prop ToyAtmo.met theta_min=1 theta_max=10 theta_step=1 (10 rays, defaults otherwise)
    "cfg1": ("GeoAc2D", ["theta_min=1", "theta_max=10", "theta_step=1"]),
    # range-dependent Cartesian main on the synthetic 5x5 grid (tests/rngdep_data.py): -prop p loc_x.dat loc_y.dat ...
    "3drd": ("GeoAc3D.RngDep", ["theta_min=10", "theta_max=30", "theta_step=20", "phi_min=-90", "phi_max=-45", "phi_step=45",
                                "bounces=1", "WriteCaustics=True", "WriteAtmo=True", "x_src=50", "y_src=-30", "z_src=0.5"]),
    # range-dependent spherical main on the synthetic lat/lon grid: -prop g loc_lat.dat loc_lon.dat ... (default source = grid centre)
    "globalrd": ("GeoAcGlobal.RngDep", ["theta_min=10", "theta_max=30", "theta_step=20", "phi_min=-90", "phi_max=-45", "phi_step=45",
                                        "bounces=1", "WriteCaustics=True", "WriteAtmo=True", "z_src=0.5", "lon_src=1.25"]),
}


# eigenray modes of the spherical mains: (binary, option, args); the verbose stdout is kept as LOG.txt
EIG_CASES = {
    "eig_global": ("GeoAcGlobal", "-eig_search", ["lat_rcvr=30", "lon_rcvr=-2.5", "bnc_min=0", "bnc_max=1", "verbose=True"]),
    "eig_global_direct": ("GeoAcGlobal", "-eig_direct", ["lat_rcvr=30.02", "lon_rcvr=-2.52", "theta_est=8.0", "bounces=0", "verbose=True"]),
    "eig_3d": ("GeoAc3D", "-eig_search", ["x_rcvr=-250", "y_rcvr=0", "bnc_min=0", "bnc_max=1", "verbose=True"]),
    "eig_3d_direct": ("GeoAc3D", "-eig_direct", ["x_rcvr=-252", "y_rcvr=3", "theta_est=8.0", "bounces=0", "verbose=True"]),
    "eig_3drd": ("GeoAc3D.RngDep", "-eig_search", ["x_rcvr=-250", "y_rcvr=20", "theta_min=2", "theta_max=20", "bounces=0", "verbose=True"]),
    "eig_globalrd": ("GeoAcGlobal.RngDep", "-eig_search", ["lat_rcvr=31.0", "lon_rcvr=-2.65", "theta_min=2", "theta_max=20", "bounces=0", "verbose=True"]),
    # -eig_direct of the two range-dependent mains (GeoAc3D.RngDep_main.cpp:593-660, GeoAcGlobal.RngDep_main.cpp:623-693): start from an estimate
    "eig_3drd_direct": ("GeoAc3D.RngDep", "-eig_direct", ["x_rcvr=-252", "y_rcvr=23", "theta_est=4.5", "bounces=0", "verbose=True"]),
    "eig_globalrd_direct": ("GeoAcGlobal.RngDep", "-eig_direct", ["lat_rcvr=31.02", "lon_rcvr=-2.67", "theta_est=5.0", "bounces=0", "verbose=True"]),
}


# -interactive sessions: (binary, args, stdin).  Two rays each (the second shows the sticky stream precision); the files are those
# left by the LAST ray of the session, LOG.txt is the whole stdout (prompts and arrival summaries).
IA_CASES = {
    "ia_global": ("GeoAcGlobal", ["WriteCaustics=True", "lat_src=35", "lon_src=-100", "z_grnd=0.3", "rng_max=400", "freq=0.5"],
                  "12.5\n-75\n1\ny\n31\n40\n0\nn\n"),
    # CalcAmp off (6-component state, no amplitude column / line) and a ray that leaves the region ("does not return")
    "ia_global_noamp": ("GeoAcGlobal", ["CalcAmp=False"], "60\n20\n0\ny\n20\n-90\n1\nn\n"),
    "ia_3d": ("GeoAc3D", ["WriteCaustics=True", "z_src=0.4", "abs_coeff=0.5"], "10\n-60\n1\ny\n44\n135\n2\nn\n"),
    "ia_2d": ("GeoAc2D", ["WriteCaustics=True", "freq=0.2"], "6\n-80\n2\ny\n25\n70\n0\nn\n"),
    "ia_3drd": ("GeoAc3D.RngDep", ["WriteCaustics=True", "x_src=50", "y_src=-30", "z_src=0.5", "CalcAmp=False"], "12\n-80\n1\ny\n28\n45\n0\nn\n"),
    "ia_globalrd": ("GeoAcGlobal.RngDep", ["WriteCaustics=True", "z_src=0.5", "lon_src=1.25", "z_grnd=0.2"], "11\n-85\n1\ny\n27\n60\n0\nn\n"),
}


def main_ia(only):
    for name, (binary, args, stdin) in IA_CASES.items():
        if only and name not in only:
            continue
        out = os.path.join(HERE, "cli", name)
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        with tempfile.TemporaryDirectory() as td:
            if binary == "GeoAcGlobal.RngDep":
                import rngdep_data as RD
                RD.write_grid_global(td)
                inputs = ["g", "loc_lat.dat", "loc_lon.dat"]
            elif binary == "GeoAc3D.RngDep":
                import rngdep_data as RD
                RD.write_grid(td)
                inputs = ["p", "loc_x.dat", "loc_y.dat"]
            else:
                shutil.copy(os.path.join(HERE, "ToyAtmo.met"), os.path.join(td, "ToyAtmo.met"))
                inputs = ["ToyAtmo.met"]
            r = subprocess.run([os.path.join(REF, binary), "-interactive"] + inputs + args, cwd=td, check=True, stdout=subprocess.PIPE,
                               input=stdin.encode(), timeout=600)
            with open(os.path.join(out, "LOG.txt"), "wb") as fh:
                fh.write(r.stdout)
            for f in ("raypath.dat", "caustics.dat"):
                if os.path.exists(os.path.join(td, f)):
                    shutil.copy(os.path.join(td, f), os.path.join(out, f))
        with open(os.path.join(out, "ARGS"), "w") as fh:
            fh.write(binary + "\n-interactive\n" + "\n".join(args) + "\n")
        with open(os.path.join(out, "STDIN"), "w") as fh:
            fh.write(stdin)
        print(name, sorted(os.listdir(out)), sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)) // 1024, "KiB")


def main_eig(only):
    for name, (binary, opt, args) in EIG_CASES.items():
        if only and name not in only:
            continue
        out = os.path.join(HERE, "cli", name)
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        with tempfile.TemporaryDirectory() as td:
            if binary == "GeoAcGlobal.RngDep":
                import rngdep_data as RD
                RD.write_grid_global(td)
                inputs = ["g", "loc_lat.dat", "loc_lon.dat"]
            elif binary == "GeoAc3D.RngDep":
                import rngdep_data as RD
                RD.write_grid(td)
                inputs = ["p", "loc_x.dat", "loc_y.dat"]
            else:
                shutil.copy(os.path.join(HERE, "ToyAtmo.met"), os.path.join(td, "ToyAtmo.met"))
                inputs = ["ToyAtmo.met"]
            r = subprocess.run([os.path.join(REF, binary), opt] + inputs + args, cwd=td, check=True, stdout=subprocess.PIPE)
            with open(os.path.join(out, "LOG.txt"), "wb") as fh:
                fh.write(r.stdout)
            for f in sorted(os.listdir(td)):
                if f.endswith(".dat") and not f.startswith("loc_"):
                    shutil.copy(os.path.join(td, f), os.path.join(out, f))
        with open(os.path.join(out, "ARGS"), "w") as fh:
            fh.write(binary + "\n" + opt + "\n" + "\n".join(args) + "\n")
        print(name, sorted(os.listdir(out)), sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)) // 1024, "KiB")


def main():
    only = sys.argv[1:]
    if only and all(o.startswith("eig") for o in only):
        return main_eig(only)
    if only and all(o.startswith("ia_") for o in only):
        return main_ia(only)
    if not only:
        main_eig(only)
        main_ia(only)
    for name, (binary, args) in CASES.items():
        if only and name not in only:
            continue
        out = os.path.join(HERE, "cli", name)
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        with tempfile.TemporaryDirectory() as td:
            if binary == "GeoAcGlobal.RngDep":
                import rngdep_data as RD
                RD.write_grid_global(td)
                inputs = ["g", "loc_lat.dat", "loc_lon.dat"]
            elif binary.endswith("RngDep"):
                import rngdep_data as RD
                RD.write_grid(td)
                inputs = ["p", "loc_x.dat", "loc_y.dat"]
            else:
                shutil.copy(os.path.join(HERE, "ToyAtmo.met"), os.path.join(td, "ToyAtmo.met"))
                inputs = ["ToyAtmo.met"]
            subprocess.run([os.path.join(REF, binary), "-prop"] + inputs + args, cwd=td, check=True,
                           stdout=subprocess.DEVNULL)
            for f in sorted(os.listdir(td)):
                if f.endswith(".dat") and not f.startswith("loc_"):
                    shutil.copy(os.path.join(td, f), os.path.join(out, f))
        with open(os.path.join(out, "ARGS"), "w") as fh:
            fh.write(binary + "\n" + "\n".join(args) + "\n")
        print(name, sorted(os.listdir(out)), sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)) // 1024, "KiB")


if __name__ == "__main__":
    main()
