"""sl.ManipulationSim (reference src/manipulation_sim.cpp:28-93, python/src/py_manipulation_sim.cpp):
a world-anchored D6 joint drags the manipulator object towards a goal pose with a linear spring
drive (stiffness 600, damping 0.1, force limit 60) while its rotation is locked; `step` sets the
drive target and advances the whole scene by one `dt`."""
from . import _math as M
from ._context import require_context
from ._math import as_mat4, f32


class ManipulationSim:
    def __init__(self, scene, manipulator, initial_pose):
        require_context()
        self._scene = scene
        self._manipulator = manipulator
        self._initial_pose = as_mat4(initial_pose)
        if manipulator not in scene._objects:   # manipulation_sim.cpp:35-37
            scene.add_object(manipulator)
        manipulator._pose = self._initial_pose.copy()
        scene.load_physics()
        manipulator._drive = {
            "flags": 1 | 2 | 4 | 8,              # drive + all three rotation axes locked (:52)
            "target": self._initial_pose[:3, 3].copy(),
            "frame": M.matrix_to_quat(self._initial_pose[:3, :3]),
            "stiffness": f32(600.0), "damping": f32(0.1), "force_limit": f32(60.0),   # :55
        }

    def set_spring_parameters(self, stiffness, damping, force_limit):
        d = self._manipulator._drive
        d["stiffness"], d["damping"], d["force_limit"] = f32(stiffness), f32(damping), f32(force_limit)

    def lock_rotation_axes(self, x, y, z):
        d = self._manipulator._drive
        d["flags"] = 1 | (2 if x else 0) | (4 if y else 0) | (8 if z else 0)

    def step(self, goal_pose, dt):
        """Drive target = initialPose^-1 * goal in the joint frame == the goal position in the
        world (manipulation_sim.cpp:83-93); one simulate(dt) of the scene (no table)."""
        from . import physics

        goal = as_mat4(goal_pose)
        self._manipulator._drive["target"] = goal[:3, 3].copy()
        physics.step_scene(self._scene, (False, 0.0), tabletop=False, dt=float(dt), frames=1, substeps=1)
