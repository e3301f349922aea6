"""Robot description loader (stdlib only).

The reference reads URDF through ``urdf_parser_py`` (``optas/models.py:12,288-290``) and only ever
uses: the robot name, the *document-ordered* joint list (name, type, parent, child, origin xyz/rpy,
axis, limit), the link list (names; inertials for RNEA), ``get_root()`` and
``get_chain(root, tip, links=False)``.  This module provides exactly that surface with
``xml.etree`` so that neither ``urdf_parser_py`` nor ``xacro`` is needed, plus a compact JSON
"kinematic constants" format (``*.kin.json``) for the robots shipped with this package.
"""
from __future__ import annotations

import json
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional


@dataclass
class Limit:
    lower: float = 0.0
    upper: float = 0.0
    velocity: float = 0.0
    effort: float = 0.0


@dataclass
class Inertial:
    mass: float
    xyz: List[float]
    rpy: List[float]
    inertia: List[float]  # ixx ixy ixz iyy iyz izz


@dataclass
class Joint:
    name: str
    type: str
    parent: str
    child: str
    xyz: Optional[List[float]] = None  # None <=> no <origin> element
    rpy: Optional[List[float]] = None
    axis: Optional[List[float]] = None  # None <=> no <axis> element
    limit: Optional[Limit] = None


@dataclass
class Link:
    name: str
    inertial: Optional[Inertial] = None


def _floats(text: Optional[str], n: int, default: float = 0.0) -> List[float]:
    if text is None:
        return [default] * n
    vals = [float(s) for s in text.split()]
    if len(vals) != n:
        raise ValueError(f"expected {n} numbers, got {text!r}")
    return vals


@dataclass
class RobotDescription:
    """Kinematic tree in URDF document order."""

    name: str
    links: List[Link] = field(default_factory=list)
    joints: List[Joint] = field(default_factory=list)

    # -- lookups ---------------------------------------------------------------------------
    @property
    def joint_map(self) -> Dict[str, Joint]:
        return {j.name: j for j in self.joints}

    @property
    def link_map(self) -> Dict[str, Link]:
        return {l.name: l for l in self.links}

    def get_root(self) -> str:
        """The unique link that is nobody's child."""
        children = {j.child for j in self.joints}
        roots = [l.name for l in self.links if l.name not in children]
        if len(roots) != 1:
            raise ValueError(f"robot '{self.name}' must have exactly one root link, found {roots}")
        return roots[0]

    def get_chain(self, root: str, tip: str, links: bool = False, joints: bool = True) -> List[str]:
        """Names from ``root`` down to ``tip`` (joints only by default, like the reference call
        ``urdf.get_chain(root, link, links=False)`` at ``optas/models.py:846``)."""
        parent_of = {j.child: j for j in self.joints}
        chain: List[str] = []
        if links:
            chain.append(tip)
        link = tip
        while link != root:
            if link not in parent_of:
                raise ValueError(f"link '{tip}' is not a descendant of '{root}'")
            j = parent_of[link]
            if joints:
                chain.append(j.name)
            link = j.parent
            if links:
                chain.append(link)
        chain.reverse()
        return chain

    # -- mutation used by RobotModel.add_base_frame ----------------------------------------------
    def add_link(self, link: Link) -> None:
        self.links.append(link)

    def add_joint(self, joint: Joint) -> None:
        self.joints.append(joint)

    # -- (de)serialisation -----------------------------------------------------------------------
    def to_dict(self) -> dict:
        def jd(j: Joint) -> dict:
            d = {"name": j.name, "type": j.type, "parent": j.parent, "child": j.child}
            if j.xyz is not None:
                d["xyz"] = j.xyz
                d["rpy"] = j.rpy
            if j.axis is not None:
                d["axis"] = j.axis
            if j.limit is not None:
                d["limit"] = {
                    "lower": j.limit.lower,
                    "upper": j.limit.upper,
                    "velocity": j.limit.velocity,
                    "effort": j.limit.effort,
                }
            return d

        def ld(l: Link) -> dict:
            d = {"name": l.name}
            if l.inertial is not None:
                d["inertial"] = {
                    "mass": l.inertial.mass,
                    "xyz": l.inertial.xyz,
                    "rpy": l.inertial.rpy,
                    "inertia": l.inertial.inertia,
                }
            return d

        return {
            "format": "optas_amd.kin/1",
            "name": self.name,
            "links": [ld(l) for l in self.links],
            "joints": [jd(j) for j in self.joints],
        }

    @staticmethod
    def from_dict(d: dict) -> "RobotDescription":
        if d.get("format") != "optas_amd.kin/1":
            raise ValueError("not an optas_amd kinematic-constants file")
        links = []
        for l in d["links"]:
            ine = l.get("inertial")
            links.append(
                Link(
                    l["name"],
                    Inertial(ine["mass"], ine["xyz"], ine["rpy"], ine["inertia"]) if ine else None,
                )
            )
        joints = []
        for j in d["joints"]:
            lim = j.get("limit")
            joints.append(
                Joint(
                    j["name"],
                    j["type"],
                    j["parent"],
                    j["child"],
                    j.get("xyz"),
                    j.get("rpy"),
                    j.get("axis"),
                    Limit(lim["lower"], lim["upper"], lim["velocity"], lim["effort"]) if lim else None,
                )
            )
        return RobotDescription(d["name"], links, joints)

    @staticmethod
    def from_json_file(filename: str) -> "RobotDescription":
        with open(filename, "r") as fh:
            return RobotDescription.from_dict(json.load(fh))

    @staticmethod
    def from_xml_string(xml: str) -> "RobotDescription":
        root = ET.fromstring(xml)
        if root.tag != "robot":
            raise ValueError("URDF root element must be <robot>")
        robot = RobotDescription(root.attrib.get("name", "robot"))
        # only *direct* children: <transmission>/<ros2_control> blocks also hold <joint name=…/> stubs
        for el in root:
            if el.tag == "link":
                ine_el = el.find("inertial")
                ine = None
                if ine_el is not None:
                    o = ine_el.find("origin")
                    m = ine_el.find("mass")
                    i = ine_el.find("inertia")
                    ine = Inertial(
                        float(m.attrib["value"]) if m is not None else 0.0,
                        _floats(o.attrib.get("xyz") if o is not None else None, 3),
                        _floats(o.attrib.get("rpy") if o is not None else None, 3),
                        [float(i.attrib.get(k, 0.0)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")]
                        if i is not None
                        else [0.0] * 6,
                    )
                robot.links.append(Link(el.attrib["name"], ine))
            elif el.tag == "joint":
                if "type" not in el.attrib:
                    continue
                o = el.find("origin")
                a = el.find("axis")
                lim = el.find("limit")
                j = Joint(
                    name=el.attrib["name"],
                    type=el.attrib["type"],
                    parent=el.find("parent").attrib["link"],
                    child=el.find("child").attrib["link"],
                )
                if o is not None:
                    j.xyz = _floats(o.attrib.get("xyz"), 3)
                    j.rpy = _floats(o.attrib.get("rpy"), 3)
                if a is not None:
                    j.axis = _floats(a.attrib.get("xyz"), 3)
                if lim is not None:
                    j.limit = Limit(
                        float(lim.attrib.get("lower", 0.0)),
                        float(lim.attrib.get("upper", 0.0)),
                        float(lim.attrib.get("velocity", 0.0)),
                        float(lim.attrib.get("effort", 0.0)),
                    )
                robot.joints.append(j)
        return robot

    @staticmethod
    def from_xml_file(filename: str) -> "RobotDescription":
        with open(filename, "r") as fh:
            return RobotDescription.from_xml_string(fh.read())


def load_robot_description(filename: str) -> RobotDescription:
    """Load ``*.urdf`` (XML) or ``*.kin.json``."""
    if filename.endswith(".json"):
        return RobotDescription.from_json_file(filename)
    return RobotDescription.from_xml_file(filename)
