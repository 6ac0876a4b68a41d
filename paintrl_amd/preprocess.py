"""Command-line part pre-processor: URDF (+ OBJ/MTL/texture) -> static tables on disk.

    python -m paintrl_amd.preprocess <part.urdf> <out.npz> [--collision hull|trimesh] [--obs-grad 4]
    python -m paintrl_amd.preprocess --synthetic door_test <out.npz>

Restates what bullet_paint_wrapper.load_part (bpw:1327-1335) computes once per process (15-37 s in the
reference) as a file that BatchedPaintEnv / PaintGymEnv can load in milliseconds.  The UV unwrap of a
raw OBJ (obj_surface_process/process_script.py, Blender) is not part of this tool.
"""
import argparse
import json
import time

from . import part_tables, synth_parts


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('urdf', nargs='?')
    ap.add_argument('out')
    ap.add_argument('--synthetic', choices=sorted(synth_parts.PARTS))
    ap.add_argument('--collision', default='hull', choices=['hull', 'trimesh'])
    ap.add_argument('--obs-grad', type=int, default=4)
    args = ap.parse_args(argv)
    t0 = time.time()
    if args.synthetic:
        tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh(args.synthetic), tex_size=(240, 240),
                                               name=args.synthetic, collision_mode=args.collision,
                                               obs_grad=args.obs_grad)
    else:
        if not args.urdf:
            ap.error('give a URDF path or --synthetic NAME')
        tables = part_tables.build_part_tables(args.urdf, collision_mode=args.collision, obs_grad=args.obs_grad)
    part_tables.save_tables(tables, args.out)
    info = tables.summary()
    info.update(seconds=round(time.time() - t0, 2), out=args.out, vertices_mutated=len(tables.vertices_mutated),
                normals_hull_corrected=int(tables.n_hull_corrected), normals_smoothed=int(tables.n_smoothed))
    print(json.dumps(info))
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
