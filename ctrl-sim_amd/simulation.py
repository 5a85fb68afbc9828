"""Device-backed stand-in for the slice of the pybind module `nocturne_cpp` that the rollout touches
(nocturne/pybind11/src/simulation.cc:20-38, scenario.cc:25-45,77-86, object.cc:33-99, vehicle.cc:19-21, road.cc:16-36,
stop_sign.cc:21):

    sim = Simulation(scenario_path, config)   # a Nocturne scenario JSON + the `scenario` config dict (cfgs/config.yaml:49-63),
    sim = Simulation(scenario)                # or a ctrlsim_amd.scenarios.Scenario (synthetic scenes)
    scn = sim.getScenario() / sim.scenario(); scn.vehicles() / getVehicles() / objects() / getObjectsThatMoved() / moving_objects()
    scn.getRoadLines() / road_lines() -> RoadLine(road_type, check_collision, geometry_points()); scn.stop_signs() -> StopSign(position())
    veh.getID() / id / getPosition() / getHeading() / getSpeed() / velocity() / getWidth() / getLength() / getGoalPosition() / getType()
    veh.position / heading / speed / collided / collision_type_veh / collision_type_edge / target_position / target_heading / target_speed
    veh.acceleration = a ; veh.brake(b) ; veh.steering = s ; veh.setPosition(x, y) | setPosition(vec) | set_position(x, y)
    veh.expert_control = False ; veh.physics_simulated = True
    sim.step(dt) ; sim.reset()

State lives on the GPU ([1, N, ...] arrays of include/ctrlsim.h); one `step` is one ctrlsim_sim_step launch (preceded by
ctrlsim_sim_set_position when a vehicle was moved) plus one small device->host read of the new state row (the evaluator reads
every vehicle every step anyway).  `veh.expert_control = True` (object.cc:58-59) is acted on as Scenario::Step does
(nocturne/cpp/src/scenario.cc:272-284): the physics step moves every body, then the vehicle is put on the logged position, heading
and speed of the new time step (ctrlsim_sim_step_expert) — what utils/sim.py:20-65 get_ground_truth_states relies on.  It needs the log
of a scenario FILE; a synthetic Scenario has none (ValueError).  Not provided: rendering, visible-state features."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib


class CollisionType:       # nocturne/pybind11/src/object.cc:19-23
    NOT_COLLIDED = 0
    VEHICLE_VEHICLE = 1
    VEHICLE_ROAD = 2


class _Enum(int):           # pybind enums: int(value) and .value both work (utils/data.py:306-307,326-328 use either)
    @property
    def value(self):
        return int(self)


class ObjectType:          # object.cc:25-31
    UNSET, VEHICLE, PEDESTRIAN, CYCLIST, OTHER = (_Enum(i) for i in range(5))


class RoadType:            # road.cc:17-26
    NONE, LANE, ROAD_LINE, ROAD_EDGE, STOP_SIGN, CROSSWALK, SPEED_BUMP, OTHER = (_Enum(i) for i in range(8))


class _Vec(SimpleNamespace):
    pass


# road type strings of the scenario files / utils/data.py:306-324 <-> RoadType
_ROAD_ENUM = {"none": 0, "lane": 1, "road_line": 2, "road_edge": 3, "stop_sign": 4, "crosswalk": 5, "speed_bump": 6, "other": 7}


class RoadLine:
    """road.cc:28-35: road_type, check_collision (= road edge, scenario.cc:1015), geometry_points()."""

    def __init__(self, kind, pts):
        self.road_type = _Enum(_ROAD_ENUM.get(kind, 7))
        self.check_collision = kind == "road_edge"
        self._pts = [_Vec(x=float(x), y=float(y)) for x, y in pts]

    def geometry_points(self):
        return self._pts

    getGeometry = geometry_points

    def canCollide(self):
        return self.check_collision


class StopSign:
    def __init__(self, x, y):
        self._p = _Vec(x=float(x), y=float(y))

    def position(self):
        return self._p


class Vehicle:
    def __init__(self, sim, i):
        self._sim, self._i = sim, i
        self.expert_control = False
        self.physics_simulated = True

    def getID(self):
        return int(self._sim.ids[self._i])       # the loader's object id (scenario.cc:894,992-997), the index for synthetic scenes

    id = property(getID)

    def getWidth(self):
        return float(self._sim.scn.width[self._i])

    def getLength(self):
        return float(self._sim.scn.length[self._i])

    width = property(getWidth)
    length = property(getLength)

    def getType(self):
        t = self._sim.scn.types[self._i]
        return _Enum(int(np.argmax(t)) if len(t) == 5 else 1)

    type = property(getType)

    def getGoalPosition(self):
        return self.target_position

    @property
    def collided(self):
        return bool(self._sim.coll_now[self._i].any())

    getCollided = lambda self: self.collided

    def _row(self):
        return self._sim.row[self._i]

    def getPosition(self):
        r = self._row()
        return _Vec(x=r[0], y=r[1])

    position = property(getPosition)

    def getHeading(self):
        return self._row()[4]

    heading = property(getHeading)

    def getSpeed(self):
        r = self._row()
        return np.float32(np.sqrt(np.float32(r[2] * r[2] + r[3] * r[3]))) if self._sim.speed is None else self._sim.speed[self._i]

    speed = property(getSpeed)

    def velocity(self):
        r = self._row()
        return _Vec(x=r[2], y=r[3])

    @property
    def collision_type_veh(self):
        return CollisionType.VEHICLE_VEHICLE if self._sim.coll_now[self._i, 0] else CollisionType.NOT_COLLIDED

    @property
    def collision_type_edge(self):
        return CollisionType.VEHICLE_ROAD if self._sim.coll_now[self._i, 1] else CollisionType.NOT_COLLIDED

    @property
    def target_position(self):
        g = self._sim.scn.goal_pos[self._i]
        return _Vec(x=g[0], y=g[1])

    @property
    def target_heading(self):
        return self._sim.scn.goal_heading[self._i]

    @property
    def target_speed(self):
        return self._sim.scn.goal_speed[self._i]

    # ---- setters (vehicle.cc:75-135): staged on the host, applied by the next sim.step()
    def setPosition(self, x, y=None):
        """Object::set_position (object.cc:52-54,87-90 -> vehicle.cc:75-87 -> b2Body::SetTransform at the current angle): staged and
        applied by the next step (ctrlsim_sim_set_position).  The rollout's own use — parking a vehicle that ran out of actions
        at (-1e6, -1e6) every step (autoregressive_policy.py:260-263) — is the device step's `exists` = 0 path."""
        if y is None:
            x, y = x.x, x.y                   # the Vector2D overload
        x, y = float(x), float(y)
        i = self._i
        if x == -1000000.0 and y == -1000000.0:
            self._sim.alive[i] = 0
            self._sim.tele.pop(i, None)       # a parked vehicle takes no earlier set_position request of this step with it
        else:
            self._sim.alive[i] = 1
            self._sim.tele[i] = (x, y)
        self._sim.row[i, 0], self._sim.row[i, 1] = np.float32(x), np.float32(y)   # Object::position_ changes at once

    set_position = setPosition

    @property
    def acceleration(self):
        return self._sim.act[self._i, 0]

    @acceleration.setter
    def acceleration(self, v):
        self._sim.act[self._i, 0] = float(v)          # Throttle(v): applied when v > 0

    def brake(self, v):
        self._sim.act[self._i, 0] = -abs(float(v))    # Brake(|v|)

    @property
    def steering(self):
        return self._sim.act[self._i, 1]

    @steering.setter
    def steering(self, v):
        self._sim.act[self._i, 1] = float(v)


class _ScenarioView:
    def __init__(self, sim):
        self._sim = sim

    def vehicles(self):
        return self._sim.vehs

    getVehicles = vehicles
    objects = vehicles                  # only vehicles are spawned here (allow_non_vehicles: cfgs/config.yaml:53)
    getObjects = vehicles

    def moving_objects(self):
        """scenario.cc:947-953: objects that exceed the speed threshold or sit away from their goal at some valid step."""
        return [v for v, m in zip(self._sim.vehs, self._sim.moving) if m]

    getObjectsThatMoved = moving_objects

    def road_lines(self):
        return self._sim.road_lines

    getRoadLines = road_lines

    def stop_signs(self):
        return self._sim.stop_sign_list

    @property
    def name(self):
        return self._sim.name


class Simulation:
    def __init__(self, scenario_path="", config=None, device="cuda:0", steps=90, dt=0.1):
        """simulation.cc:20-27 / scenario.h:75-103: (scenario_path, config).  scenario_path: a Nocturne scenario JSON (path, file
        object or parsed dict) read by ctrlsim_amd.ingest.load_nocturne_json with the `scenario` config keys the C++ loader reads
        (start_time is REQUIRED like config.at("start_time"); allow_non_vehicles defaults to True, spawn_invalid_objects to False,
        moving_threshold 0.2, speed_threshold 0.05) — or, for synthetic scenes, a ctrlsim_amd.scenarios.Scenario.  An empty path
        raises ValueError (std::invalid_argument "No scenario file inputted.", scenario.h:99-103)."""
        from .scenarios import Scenario
        config = dict(config or {})
        self.moving, self.ids, self.name, road_data = None, None, "", None
        if isinstance(scenario_path, Scenario):
            scenario = scenario_path
        else:
            if isinstance(scenario_path, str) and not scenario_path:
                raise ValueError("No scenario file inputted.")
            from . import ingest
            scenario, info = ingest.load_nocturne_json(
                scenario_path, start_time=int(config["start_time"]), allow_non_vehicles=bool(config.get("allow_non_vehicles", True)),
                spawn_invalid_objects=bool(config.get("spawn_invalid_objects", False)), steps=steps,
                moving_threshold=float(config.get("moving_threshold", 0.2)), speed_threshold=float(config.get("speed_threshold", 0.05)))
            self.moving, self.ids, self.name, road_data = info["moving"], info["ids"], info["name"], info["road_data"]
            self.gt_data_dict = info["gt_data_dict"]
        self.scn = scenario
        self.device = torch.device(device)
        self.lib = _lib.lib()
        self.N = scenario.N
        self.steps = steps
        self.dt = dt
        if self.moving is None:
            self.moving, self.ids = np.ones(self.N, bool), np.arange(self.N)
        # road lines / stop signs: the file's polylines when there is a file, else the synthetic scene's chunks (one line per chunk)
        if road_data is None:
            from .scenarios import ROAD_TYPES
            inv = {v: k for k, v in ROAD_TYPES.items()}
            road_data = []
            for pl, ty in zip(scenario.road_points, scenario.road_types):
                n = int(pl[:, 2].sum())
                road_data.append({"geometry": [{"x": float(q[0]), "y": float(q[1])} for q in pl[:n]], "type": inv[int(np.argmax(ty))]})
        self.road_lines = [RoadLine(r["type"], [(q["x"], q["y"]) for q in r["geometry"]]) for r in road_data
                           if not isinstance(r["geometry"], dict)]
        self.stop_sign_list = [StopSign(r["geometry"]["x"], r["geometry"]["y"]) for r in road_data if isinstance(r["geometry"], dict)]
        dev = self.device
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        s = scenario
        self.E = max(1, len(s.edge_segments))
        edges = np.full((1, self.E, 4), 1e30, np.float32)
        edges[0, :len(s.edge_segments)] = s.edge_segments
        self.init_pose = f32(np.stack([s.x, s.y, s.heading, s.speed], 1)[None])
        self.size = f32(np.stack([s.length, s.width], 1)[None])
        self.edges = f32(edges)
        # the step receives continuous (accel, steer) pairs, so the action grid is not used; the values are the reference's
        # (cfgs/dataset/waymo/base.yaml:13-16,41-42) for callers that pass token ids through the same entry point
        self.disc6 = (C.c_double * 6)(-10, 10, -0.7, 0.7, 20, 50)
        self.vehs = [Vehicle(self, i) for i in range(self.N)]
        self.guard = torch.zeros(2, dtype=torch.int32, device=dev)   # guard pair (ctrlsim_bind): [1] = simulator events
        self.speed = None
        self.reset()

    def scenario(self):
        return _ScenarioView(self)

    def getScenario(self):
        return _ScenarioView(self)

    def reset(self):
        dev, N, T1 = self.device, self.N, self.steps + 1
        for v in self.vehs:                                 # simulation.cc:29-38: reset() loads the scenario again — fresh objects
            v.expert_control, v.physics_simulated = False, True
        self.exists = torch.ones(1, N, dtype=torch.uint8, device=dev)
        self.phys = torch.zeros(1, N, 20, device=dev)
        self.hist = torch.zeros(1, N, T1, 8, device=dev)
        self.coll = torch.zeros(1, N, T1, 2, dtype=torch.uint8, device=dev)
        self.t = 0
        self.log_t = 0                                      # index into the expert logs (scenario.cc:267: current_time_ - start_time)
        self.alive = np.ones(N, np.uint8)
        self.tele = {}                                      # vehicle -> (x, y): set_position requests of this step
        self.act = np.zeros((N, 2), np.float64)
        p = _lib.ptr
        self.contact_state = torch.zeros(1, int(self.lib.ctrlsim_sim_contact_floats(N)), device=self.device)
        _lib.check(self.lib.ctrlsim_sim_init(1, N, self.E, p(self.init_pose), p(self.size), p(self.edges), p(self.exists),
                                             p(self.phys), p(self.hist), p(self.coll), T1, p(self.contact_state),
                                             _lib.stream_ptr()), "sim_init")
        self._read()

    def _read(self):
        torch.cuda.synchronize(self.device)
        self.row = self.hist[0, :, self.t].cpu().numpy()
        self.coll_now = self.coll[0, :, self.t].cpu().numpy()
        self.speed = self.phys[0, :, 16].cpu().numpy()

    def step(self, dt):
        if self.t >= self.steps:
            raise RuntimeError("rollout longer than the allocated history")
        p = _lib.ptr
        # Scenario::Step, scenario.cc:267: current_time_ += static_cast<int>(dt / 0.1) — the log index of the NEW time step
        log_t = self.log_t + int(float(np.float32(dt)) / 0.1)
        expert = None
        if any(v.expert_control for v in self.vehs):
            # scenario.cc:276-283: expert_trajectories_ / _headings_ / _speeds_ .at(id).at(current_time_) of the new time.  Validated
            # BEFORE anything is bound or launched: a raise must leave neither a dangling guard binding nor a half-applied step
            if getattr(self, "gt_data_dict", None) is None:
                raise ValueError("expert_control needs the logged trajectories of a scenario file (Simulation(scenario_path, config))")
            ex = np.full((1, self.N, 4), np.nan, np.float32)
            for i, v in enumerate(self.vehs):
                if v.expert_control:
                    g = self.gt_data_dict[int(self.ids[i])]
                    tr = g["traj"]
                    if log_t >= min(len(tr), int(g.get("log_len", len(tr)))):
                        raise IndexError("expert trajectory shorter than the rollout (std::out_of_range in the reference, scenario.cc:280)")
                    ex[0, i] = tr[log_t, :4]
            expert = torch.from_numpy(ex).to(self.device)
        # the step's guard events (contacts beyond the island solver's table) go to THIS simulation's pair, not to whichever engine bound
        # its own last (ctrlsim_bind: per-caller state); the binding found here is put back afterwards, whatever happens in between — an
        # engine that bound once for a whole run keeps receiving its events, and the library never keeps a pointer to a freed tensor
        prev = self.lib.ctrlsim_bound_guard()
        _lib.check(self.lib.ctrlsim_bind(-1, p(self.guard)), "bind")
        try:
            self.exists.copy_(torch.from_numpy(self.alive[None]).to(self.device))
            if self.tele:
                xy = np.full((1, self.N, 2), np.nan, np.float32)
                for i, q in self.tele.items():
                    xy[0, i] = q
                _lib.check(self.lib.ctrlsim_sim_set_position(1, self.N, p(torch.from_numpy(xy).to(self.device)), p(self.phys),
                                                             _lib.stream_ptr()), "sim_set_position")
                self.tele = {}
            act = torch.from_numpy(self.act[None].copy()).to(self.device)
            if expert is None:
                _lib.check(self.lib.ctrlsim_sim_step(1, self.N, self.E, None, p(act), self.disc6, p(self.size), p(self.edges),
                                                     p(self.exists), p(self.phys), p(self.hist), p(self.coll), None, self.t,
                                                     self.steps + 1, float(dt), 0, p(self.contact_state), _lib.stream_ptr()), "sim_step")
            else:
                _lib.check(self.lib.ctrlsim_sim_step_expert(1, self.N, self.E, None, p(act), self.disc6, p(self.size), p(self.edges),
                                                            p(self.exists), p(self.phys), p(self.hist), p(self.coll), None, self.t,
                                                            self.steps + 1, float(dt), p(self.contact_state), p(expert),
                                                            _lib.stream_ptr()), "sim_step_expert")
            self.t += 1
            self.log_t = log_t
            self._read()                                        # synchronises: the step's events are in self.guard now
        finally:
            self.lib.ctrlsim_bind(-1, prev)
        n = int(self.guard[1].item())
        if n or int(self.guard[0].item()):
            self.guard.zero_()
        if n:
            raise FloatingPointError(f"{n} simulator contacts beyond the island solver's table in this step (csrc/sim.hip: MAX_ISLAND_CONTACTS)")
