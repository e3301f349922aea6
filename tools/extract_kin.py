"""Convert a URDF into the compact kinematic-constants JSON shipped with optas_amd.

Usage (run in the build container, where the reference checkout is mounted):
    python tools/extract_kin.py /root/reference/example/robots/kuka_lwr/kuka_lwr.urdf optas_amd/robots/kuka_lwr.kin.json

Only numbers that the hot path consumes are kept (joint tree, origins, axes, limits, link
inertials); meshes, visuals, collisions, transmissions and gazebo tags are dropped.  The output is
data, not code: nothing under /root/reference is needed once the JSON exists.
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from optas_amd.urdf import RobotDescription  # noqa: E402


def main(src: str, dst: str) -> None:
    robot = RobotDescription.from_xml_file(src)
    d = robot.to_dict()
    d["source"] = os.path.basename(src)
    with open(dst, "w") as fh:
        json.dump(d, fh, indent=1)
    print(f"{src} -> {dst}: {len(robot.joints)} joints, {len(robot.links)} links, root={robot.get_root()}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
