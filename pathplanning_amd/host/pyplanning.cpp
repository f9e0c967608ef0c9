// pybind11 module `pyplanning`: the reference's Python surface (interfaces/python/src/pyplanning.cpp), same class / method names, backed
// by libpphip.so through planner_hip.hpp.  Every name the reference's module binds is bound here (tests/golden/pyplanning_bound_names.json,
// tests/test_pyplanning_surface.py): initialize, Status, Point2d, Pose2d, GridCellPosition, Steer, Direction, the path value types and
// connections, KinematicBicycleModel, StateSpaceSE2, OccupancyMap (+ set_grids), ObstacleListOccupancyMap, Obstacle and the shapes, GVD,
// StateValidatorSE2Base / SE2Free / OccupancyMap, HybridAStarSearchParameters / SmootherParameters / Stats, PathPlannerSE2Base, HybridAStar,
// the N2 heuristics / propagators / planners.  New surface next to it: search_batch, GridAStarBatch, RRT / RRTStar (the reference does not
// bind RRT).
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/operators.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "planner_hip.hpp"
#include "a_star.hpp"
#include "map_authoring.hpp"

namespace py = pybind11;
using namespace Planner;

PYBIND11_MODULE(pyplanning, m)
{
	m.def("initialize", []() {}); // PP_INIT: logging / profiler singletons of the reference; nothing to set up here

	py::enum_<Status>(m, "Status").value("SUCCESS", Status::Success).value("FAILURE", Status::Failure);
	py::enum_<Steer>(m, "Steer").value("LEFT", Steer::Left).value("STRAIGHT", Steer::Straight).value("RIGHT", Steer::Right);
	py::enum_<Direction>(m, "Direction").value("FORWARD", Direction::Forward).value("BACKWARD", Direction::Backward).value("NO_MOTION", Direction::NoMotion);

	py::class_<Point2d>(m, "Point2d")
		.def(py::init<double, double>())
		.def("x", [](Point2d& p) { return p.x(); })
		.def("y", [](Point2d& p) { return p.y(); })
		.def(py::self + py::self)
		.def(py::self - py::self)
		.def(py::self == py::self)
		.def(py::self != py::self);

	py::class_<Pose2d>(m, "Pose2d")
		.def(py::init<const Point2d&, double>())
		.def(py::init<double, double, double>())
		.def_readwrite("position", &Pose2d::position)
		.def_readwrite("theta", &Pose2d::theta)
		.def("x", [](Pose2d& p) { return p.x(); })
		.def("y", [](Pose2d& p) { return p.y(); })
		.def(py::self + py::self)
		.def(py::self - py::self)
		.def(py::self == py::self)
		.def(py::self != py::self);

	// ---- paths (pyplanning.cpp:247-309) ----
	struct PathSE2BaseWrapper : PathSE2Base {
		using PathSE2Base::PathSE2Base;
		Pose2d Interpolate(double ratio) const override { PYBIND11_OVERRIDE_PURE(Pose2d, PathSE2Base, Interpolate, ratio); }
		void Truncate(double ratio) override { PYBIND11_OVERRIDE_PURE(void, PathSE2Base, Truncate, ratio); }
	};
	py::class_<PathSE2Base, Ref<PathSE2Base>, PathSE2BaseWrapper>(m, "PathSE2Base")
		.def(py::init<>())
		.def(py::init<Pose2d, double>(), py::arg("init"), py::arg("length") = 0.0)
		.def("get_initial_state", &PathSE2Base::GetInitialState)
		.def("get_final_state", &PathSE2Base::GetFinalState)
		.def("interpolate", py::overload_cast<double>(&PathSE2Base::Interpolate, py::const_))
		.def("interpolate", py::overload_cast<const std::vector<double>&>(&PathSE2Base::Interpolate, py::const_))
		.def("truncate", &PathSE2Base::Truncate)
		.def("get_length", &PathSE2Base::GetLength);
	py::class_<PathSE2, Ref<PathSE2>, PathSE2Base>(m, "PathSE2").def(py::init<const Pose2d&, const Pose2d&>());
	py::class_<PathNonHolonomicSE2Base, Ref<PathNonHolonomicSE2Base>, PathSE2Base>(m, "PathNonHolonomicSE2Base")
		.def("get_direction", &PathNonHolonomicSE2Base::GetDirection)
		.def("get_cusp_point_ratios", &PathNonHolonomicSE2Base::GetCuspPointRatios);
	py::class_<PathReedsShepp, Ref<PathReedsShepp>, PathNonHolonomicSE2Base>(m, "PathReedsShepp")
		.def("interpolate", py::overload_cast<double>(&PathReedsShepp::Interpolate, py::const_))
		.def("interpolate", py::overload_cast<const std::vector<double>&>(&PathReedsShepp::Interpolate, py::const_))
		.def_property("min_turning_radius", &PathReedsShepp::GetMinTurningRadius, nullptr)
		.def_property_readonly("word", [](const PathReedsShepp& p) { return p.Record().word; })
		.def_property_readonly("motions", [](const PathReedsShepp& p) {
			py::list out;
			const pp_rs_path& r = p.Record();
			for (int i = 0; i < 5; i++)
				if (r.motion_length[i] != INFINITY && r.direction[i] != 2)
					out.append(py::make_tuple((Steer)r.steer[i], (Direction)r.direction[i], r.motion_length[i]));
				else
					break;
			return out;
		});
	py::class_<KinematicBicycleModel, Ref<KinematicBicycleModel>>(m, "KinematicBicycleModel")
		.def(py::init<double, double>(), py::arg("wheelbase") = 2.6, py::arg("rear_to_center") = 0.0)
		.def("constant_steer", &KinematicBicycleModel::ConstantSteer, py::arg("from"), py::arg("steering"), py::arg("dist"), py::arg("direction") = Direction::Forward)
		.def("get_steering_angle_from_turning_radius", &KinematicBicycleModel::GetSteeringAngleFromTurningRadius);
	py::class_<PathConstantSteer, Ref<PathConstantSteer>, PathNonHolonomicSE2Base>(m, "PathConstantSteer")
		.def(py::init<const Ref<KinematicBicycleModel>&, const Pose2d&, double, double, Direction>())
		.def_property("steering", &PathConstantSteer::GetSteeringAngle, nullptr);
	struct PathConnectionSE2BaseWrapper : PathConnectionSE2Base {
		using PathConnectionSE2Base::PathConnectionSE2Base;
		Ref<PathSE2Base> Connect(const Pose2d& from, const Pose2d& to) override { PYBIND11_OVERRIDE_PURE(Ref<PathSE2Base>, PathConnectionSE2Base, Connect, from, to); }
	};
	py::class_<PathConnectionSE2Base, Ref<PathConnectionSE2Base>, PathConnectionSE2BaseWrapper>(m, "PathConnectionSE2Base")
		.def(py::init<>())
		.def("connect", &PathConnectionSE2Base::Connect);
	py::class_<PathConnectionSE2, Ref<PathConnectionSE2>, PathConnectionSE2Base>(m, "PathConnectionSE2").def(py::init<>());
	py::class_<PathConnectionReedsShepp, Ref<PathConnectionReedsShepp>, PathConnectionSE2Base>(m, "PathConnectionReedsShepp")
		.def(py::init<double, double, double, double>(), py::arg("min_turning_radius") = 1.0, py::arg("direction_switching_cost") = 0.0, py::arg("reverse_cost_multiplier") = 1.0,
			py::arg("forward_cost_multiplier") = 1.0);

	py::class_<GridCellPosition>(m, "GridCellPosition")
		.def(py::init<>())
		.def(py::init<int, int>())
		.def_readwrite("row", &GridCellPosition::row)
		.def_readwrite("col", &GridCellPosition::col)
		.def(py::self == py::self)
		.def(py::self != py::self)
		.def("__hash__", [](const GridCellPosition& c) { return std::hash<GridCellPosition>()(c); })
		.def("__repr__", [](const GridCellPosition& c) { return "<GridCellPosition: row " + std::to_string(c.row) + ", col: " + std::to_string(c.col) + ">"; });

	py::class_<StateSpaceSE2, Ref<StateSpaceSE2>>(m, "StateSpaceSE2")
		.def(py::init<const std::array<Pose2d, 2>&>())
		.def(py::init<const Pose2d&, const Pose2d&>())
		.def("enforce_bounds", &StateSpaceSE2::EnforceBounds)
		.def("validate_bounds", &StateSpaceSE2::ValidateBounds)
		.def("sample_uniform", &StateSpaceSE2::SampleUniform)   // pyplanning.cpp:325
		.def("sample_gaussian", &StateSpaceSE2::SampleGaussian) // pyplanning.cpp:326
		.def_readonly("bounds", &StateSpaceSE2::bounds);
	// not in the reference (its engine is seeded from std::random_device only): a fixed seed for the global engine the two samplers draw from
	m.def("seed_random", [](unsigned long long seed) { Random<double>::Seed(seed); });

	struct OccupancyMapWrapper : OccupancyMap { // pyplanning.cpp:337-341: Python may subclass the map
		using OccupancyMap::OccupancyMap;
		bool IsOccupied(const GridCellPosition& a) override { PYBIND11_OVERRIDE(bool, OccupancyMap, IsOccupied, a); }
	};
	py::class_<OccupancyMap, Ref<OccupancyMap>, OccupancyMapWrapper>(m, "OccupancyMap")
		.def(py::init<float>())
		.def("initialize_size", &OccupancyMap::InitializeSize)
		.def("rows", &OccupancyMap::Rows)
		.def("columns", &OccupancyMap::Columns)
		.def("set_position", &OccupancyMap::SetPosition)
		.def("get_position", &OccupancyMap::GetPosition)
		.def("update", &OccupancyMap::Update)
		.def("is_occupied", &OccupancyMap::IsOccupied)
		.def("is_occupied)", &OccupancyMap::IsOccupied) // the name the reference actually registers (pyplanning.cpp:351, a typo): reachable by getattr only
		// bound by the reference (pyplanning.cpp:354) although GVD::ObstacleDistanceMap is not a bound class: calling it there raises
		// TypeError ("Unable to convert function return value to a Python type"); the same here
		.def("get_obstacle_map", [](OccupancyMap&) -> py::object {
			throw py::type_error("Unable to convert function return value to a Python type! The signature was\n\t(self: pyplanning.OccupancyMap) -> GVD::ObstacleDistanceMap");
		})
		.def("get_occupancy_value", py::overload_cast<int, int>(&OccupancyMap::GetOccupancyValue))
		.def("get_occupancy_value", py::overload_cast<const GridCellPosition&>(&OccupancyMap::GetOccupancyValue))
		.def("grid_cell_to_local_position", &OccupancyMap::GridCellToLocalPosition)
		.def("local_position_to_grid_cell", &OccupancyMap::LocalPositionToGridCell, py::arg("position"), py::arg("bounded") = true)
		.def("local_position_to_world_position", &OccupancyMap::LocalPositionToWorldPosition)
		.def("world_position_to_local_position", &OccupancyMap::WorldPositionToLocalPosition)
		.def("occupancy", [](OccupancyMap& map) {
			py::array_t<int32_t> a({ map.Rows(), map.Columns() });
			std::copy(map.Occupancy().begin(), map.Occupancy().end(), a.mutable_data());
			return a;
		})
		.def("world_position_to_grid_cell", &OccupancyMap::WorldPositionToGridCell, py::arg("position"), py::arg("bounded") = true)
		.def("grid_cell_to_world_position", &OccupancyMap::GridCellToWorldPosition)
		.def("is_inside_map", py::overload_cast<const GridCellPosition&>(&OccupancyMap::IsInsideMap, py::const_))
		.def("is_inside_map", py::overload_cast<const Point2d&>(&OccupancyMap::IsInsideMap, py::const_))
		.def("set_grids",
			[](OccupancyMap& map, py::array_t<int32_t, py::array::c_style | py::array::forcecast> occ, py::array_t<int32_t, py::array::c_style | py::array::forcecast> d2,
				py::array_t<float, py::array::c_style | py::array::forcecast> pc) {
				const size_t n = (size_t)map.Rows() * map.Columns();
				if ((size_t)occ.size() != n || (size_t)d2.size() != n || (size_t)pc.size() != n)
					throw std::invalid_argument("set_grids: arrays must be rows x columns");
				map.SetGrids(occ.data(), d2.data(), pc.data());
			},
			py::arg("occupancy"), py::arg("dist2"), py::arg("path_cost"))
		.def("set_distances",
			[](OccupancyMap& map, py::array_t<float, py::array::c_style | py::array::forcecast> d) {
				if ((size_t)d.size() != (size_t)map.Rows() * map.Columns())
					throw std::invalid_argument("set_distances: array must be rows x columns");
				map.SetDistances(d.data());
			},
			py::arg("distance"))
		.def("get_distance_to_nearest_obstacle", &OccupancyMap::GetDistanceToNearestObstacle)
		.def("set_nearest_cells",
			[](OccupancyMap& map, py::array_t<int32_t, py::array::c_style | py::array::forcecast> o, py::array_t<int32_t, py::array::c_style | py::array::forcecast> e) {
				const size_t n = (size_t)map.Rows() * map.Columns() * 2;
				if ((size_t)o.size() != n || (size_t)e.size() != n)
					throw std::invalid_argument("set_nearest_cells: arrays must be rows x columns x 2");
				map.SetNearestCells(o.data(), e.data());
			},
			py::arg("nearest_obstacle"), py::arg("nearest_edge"));

	struct StateValidatorSE2BaseWrapper : StateValidatorSE2Base { // pyplanning.cpp:402-406 (whose IsPathValid trampoline dispatches to IsStateValid, Q18)
		using StateValidatorSE2Base::StateValidatorSE2Base;
		bool IsStateValid(const Pose2d& a) override { PYBIND11_OVERRIDE_PURE(bool, StateValidatorSE2Base, IsStateValid, a); }
		bool IsPathValid(const PathSE2Base& a, float* b) override { PYBIND11_OVERRIDE_PURE(bool, StateValidatorSE2Base, IsPathValid, a, b); }
	};
	py::class_<StateValidatorSE2Base, Ref<StateValidatorSE2Base>, StateValidatorSE2BaseWrapper>(m, "StateValidatorSE2Base")
		.def(py::init<const Ref<StateSpaceSE2>&>())
		.def("is_state_valid", &StateValidatorSE2Base::IsStateValid)
		.def("is_path_valid", [](StateValidatorSE2Base& v, const PathSE2Base& path) { return v.IsPathValid(path, (float*)nullptr); })
		.def("is_path_valid_with_ratio",
			[](StateValidatorSE2Base& v, const PathSE2Base& path) {
				float last = 0.0f;
				const bool ok = v.IsPathValid(path, &last);
				return py::make_tuple(ok, last);
			})
		.def_property("state_space", &StateValidatorSE2Base::GetStateSpace, nullptr);
	py::class_<StateValidatorSE2Free, Ref<StateValidatorSE2Free>, StateValidatorSE2Base>(m, "StateValidatorSE2Free").def(py::init<const Ref<StateSpaceSE2>&>());

	// ---- map authoring (pyplanning.cpp:364-400, 423-435) ----
	py::class_<ObstacleListOccupancyMap, Ref<ObstacleListOccupancyMap>, OccupancyMap>(m, "ObstacleListOccupancyMap")
		.def(py::init<float>())
		.def("add_obstacle", &ObstacleListOccupancyMap::AddObstacle)
		.def("remove_obstacle", &ObstacleListOccupancyMap::RemoveObstacle)
		.def("get_num_obstacles", &ObstacleListOccupancyMap::GetNumObstacles);
	py::class_<Obstacle, Ref<Obstacle>>(m, "Obstacle")
		.def(py::init<>())
		.def("set_shape", &Obstacle::SetShape)
		.def("set_pose", &Obstacle::SetPose)
		.def("get_boundary_grid_cell_position", &Obstacle::GetBoundaryGridCellPosition)
		.def("get_boundary_world_position", &Obstacle::GetBoundaryWorldPosition);
	struct ShapeWrapper : Shape {
		using Shape::Shape;
		void GetGridCellsPosition(OccupancyMap& a, const Pose2d& b, std::vector<GridCellPosition>& c) override { PYBIND11_OVERRIDE_PURE(void, Shape, GetGridCellsPosition, a, b, c); }
		void GetVerticesPosition(const Pose2d& a, std::vector<Point2d>& b) override { PYBIND11_OVERRIDE_PURE(void, Shape, GetVerticesPosition, a, b); }
	};
	py::class_<Shape, Ref<Shape>, ShapeWrapper>(m, "Shape").def(py::init<>());
	py::class_<CompositeShape, Ref<CompositeShape>, Shape>(m, "CompositeShape").def(py::init<>()).def("add", &CompositeShape::Add);
	py::class_<PolygonShape, Ref<PolygonShape>, Shape>(m, "PolygonShape").def(py::init<const std::vector<Point2d>&>());
	py::class_<RegularPolygonShape, Ref<RegularPolygonShape>, Shape>(m, "RegularPolygonShape").def(py::init<double, int>());
	py::class_<RectangleShape, Ref<RectangleShape>, Shape>(m, "RectangleShape").def(py::init<double, double>());
	py::class_<CircleShape, Ref<CircleShape>, Shape>(m, "CircleShape").def(py::init<double, int>());
	py::class_<GVD>(m, "GVD")
		.def(py::init<const Ref<OccupancyMap>&>())
		.def("update", &GVD::Update)
		// not in the reference's module: how Update builds the two distance maps ("reference_order", the default: the reference's
		// brushfire bit for bit; "exact_transform": the device's exact Euclidean transform, see pp_hip.h: pp_map_update_gvd_ex)
		.def("set_update_mode", [](GVD& g, const std::string& mode) {
			if (mode == "reference_order")
				g.SetUpdateMode(OccupancyMap::FieldUpdateMode::ReferenceOrder);
			else if (mode == "exact_transform")
				g.SetUpdateMode(OccupancyMap::FieldUpdateMode::ExactTransform);
			else
				throw std::invalid_argument("mode: 'reference_order' or 'exact_transform'");
		})
		.def("get_distance_to_nearest_obstacle", py::overload_cast<int, int>(&GVD::GetDistanceToNearestObstacle, py::const_))
		.def("get_distance_to_nearest_obstacle", py::overload_cast<const GridCellPosition&>(&GVD::GetDistanceToNearestObstacle, py::const_))
		.def("get_distance_to_nearest_voronoi_edge", py::overload_cast<int, int>(&GVD::GetDistanceToNearestVoronoiEdge, py::const_))
		.def("get_distance_to_nearest_voronoi_edge", py::overload_cast<const GridCellPosition&>(&GVD::GetDistanceToNearestVoronoiEdge, py::const_))
		.def("get_path_cost", py::overload_cast<int, int>(&GVD::GetPathCost, py::const_))
		.def("get_path_cost", py::overload_cast<const GridCellPosition&>(&GVD::GetPathCost, py::const_))
		.def("get_nearest_obstacle_cell", py::overload_cast<int, int>(&GVD::GetNearestObstacleCell, py::const_))
		.def("get_nearest_voronoi_edge_cell", py::overload_cast<int, int>(&GVD::GetNearestVoronoiEdgeCell, py::const_))
		.def("visualize", &GVD::Visualize);

	py::class_<StateValidatorOccupancyMap, Ref<StateValidatorOccupancyMap>, StateValidatorSE2Base>(m, "StateValidatorOccupancyMap")
		.def(py::init<const Ref<StateSpaceSE2>&, const Ref<OccupancyMap>&>())
		.def("get_occupancy_map", &StateValidatorOccupancyMap::GetOccupancyMap)
		.def("is_state_valid", py::overload_cast<const Pose2d&>(&StateValidatorOccupancyMap::IsStateValid))
		.def("is_states_valid",
			[](StateValidatorOccupancyMap& v, py::array_t<double, py::array::c_style | py::array::forcecast> poses) {
				if (poses.ndim() != 2 || poses.shape(1) != 3)
					throw std::invalid_argument("poses must be (n, 3)");
				py::array_t<uint8_t> out(poses.shape(0));
				if (poses.shape(0))
					ppCheck(pp_check_states(v.Device(), poses.shape(0), poses.data(), out.mutable_data()));
				return out;
			})
		.def("is_arc_valid",
			[](StateValidatorOccupancyMap& v, const Pose2d& from, double curvature, double length, Direction dir) {
				float last = 0;
				bool ok = v.IsArcValid(from, curvature, length, dir, &last);
				return py::make_tuple(ok, last);
			})
		.def_readwrite("min_path_interpolation_distance", &StateValidatorOccupancyMap::minPathInterpolationDistance)
		.def_readwrite("min_safe_radius", &StateValidatorOccupancyMap::minSafeRadius);

	struct PathPlannerSE2BaseWrapper : PathPlannerSE2Base {
		using PathPlannerSE2Base::PathPlannerSE2Base;
		Status SearchPath() override { PYBIND11_OVERRIDE_PURE(Status, PathPlannerSE2Base, SearchPath); }
		std::vector<Pose2d> GetPath() const override { PYBIND11_OVERRIDE_PURE(std::vector<Pose2d>, PathPlannerSE2Base, GetPath); }
	};
	py::class_<PathPlannerSE2Base, PathPlannerSE2BaseWrapper>(m, "PathPlannerSE2Base")
		.def(py::init<>())
		.def("search_path", &PathPlannerSE2Base::SearchPath)
		.def("get_path", &PathPlannerSE2Base::GetPath)
		.def("set_init_state", &PathPlannerSE2Base::SetInitState)
		.def("set_goal_state", &PathPlannerSE2Base::SetGoalState);

	py::class_<HybridAStar::SearchParameters>(m, "HybridAStarSearchParameters")
		.def(py::init<>())
		.def(py::init<double, double, double, double, double, unsigned int, double, double>())
		.def_readonly("wheelbase", &HybridAStar::SearchParameters::wheelbase)
		.def_readonly("min_turning_radius", &HybridAStar::SearchParameters::minTurningRadius)
		.def_readonly("direction_switching_cost", &HybridAStar::SearchParameters::directionSwitchingCost)
		.def_readonly("reverse_cost_multiplier", &HybridAStar::SearchParameters::reverseCostMultiplier)
		.def_readonly("forward_cost_multiplier", &HybridAStar::SearchParameters::forwardCostMultiplier)
		.def_readonly("voronoi_cost_multiplier", &HybridAStar::SearchParameters::voronoiCostMultiplier)
		.def_readonly("num_generated_motion", &HybridAStar::SearchParameters::numGeneratedMotion)
		.def_readonly("spatial_resolution", &HybridAStar::SearchParameters::spatialResolution)
		.def_readonly("angular_resolution", &HybridAStar::SearchParameters::angularResolution);

	py::class_<Smoother::Parameters>(m, "HybridAStarSmootherParameters") // pyplanning.cpp:86-97
		.def(py::init<float>())
		.def_readwrite("step_tolerance", &Smoother::Parameters::stepTolerance)
		.def_readwrite("max_iterations", &Smoother::Parameters::maxIterations)
		.def_readwrite("learning_rate", &Smoother::Parameters::learningRate)
		.def_readwrite("path_weight", &Smoother::Parameters::pathWeight)
		.def_readwrite("smooth_weight", &Smoother::Parameters::smoothWeight)
		.def_readwrite("voronoi_weight", &Smoother::Parameters::voronoiWeight)
		.def_readwrite("collision_weight", &Smoother::Parameters::collisionWeight)
		.def_readwrite("curvature_weight", &Smoother::Parameters::curvatureWeight)
		.def_readwrite("collision_ratio", &Smoother::Parameters::collisionRatio)
		.def_readonly("max_curvature", &Smoother::Parameters::maxCurvature);
	py::enum_<Smoother::Status>(m, "SmoothingStatus") // pyplanning.cpp:103-108
		.value("MAX_ITERATION", Smoother::Status::MaxIteration)
		.value("STEP_TOLERANCE", Smoother::Status::StepTolerance)
		.value("PATH_SIZE", Smoother::Status::PathSize)
		.value("FAILURE", Smoother::Status::Failure)
		.value("COLLISION", Smoother::Status::Collision);
	py::class_<HybridAStar::Stats>(m, "HybridAStarStats")
		.def_readonly("graph_search_status", &HybridAStar::Stats::graphSearchStatus)
		.def_readonly("smoothing_status", &HybridAStar::Stats::smoothingStatus);

	py::class_<HybridAStar, PathPlannerSE2Base>(m, "HybridAStar")
		.def(py::init<>())
		.def(py::init<const HybridAStar::SearchParameters&>())
		.def(py::init<const HybridAStar::SearchParameters&, int, int>(), py::arg("parameters"), py::arg("max_batch"), py::arg("max_nodes") = 81920)
		.def("initialize", &HybridAStar::Initialize)
		.def_readwrite("path_interpolation", &HybridAStar::pathInterpolation)
		.def("get_stats", &HybridAStar::GetStats)
		.def("get_graph_search_optimal_cost", &HybridAStar::GetGraphSearchOptimalCost)
		.def("get_graph_search_path", &HybridAStar::GetGraphSearchPath)
		.def("get_graph_search_nodes", &HybridAStar::GetGraphSearchNodes)
		.def("get_graph_search_explored_path_set", &HybridAStar::GetGraphSearchExploredPathSet)
		.def("visualize_obstacle_heuristic", &HybridAStar::VisualizeObstacleHeuristic)
		.def("get_smoothed_path", &HybridAStar::GetSmoothedPath)
		.def_property("smoother_parameters", &HybridAStar::GetSmootherParameters, &HybridAStar::SetSmootherParameters)
		.def("get_search_parameters", &HybridAStar::GetSearchParameters)
		.def("set_seed", &HybridAStar::SetSeed)
		.def("search_batch",
			[](HybridAStar& h, py::array_t<double, py::array::c_style | py::array::forcecast> starts, py::array_t<double, py::array::c_style | py::array::forcecast> goals,
				py::array_t<uint64_t, py::array::c_style | py::array::forcecast> seeds) {
				const size_t n = starts.shape(0);
				std::vector<Pose2d> s(n), g(n);
				std::vector<uint64_t> sd(seeds.data(), seeds.data() + n);
				for (size_t i = 0; i < n; i++) {
					s[i] = Pose2d(starts.at(i, 0), starts.at(i, 1), starts.at(i, 2));
					g[i] = Pose2d(goals.at(i, 0), goals.at(i, 1), goals.at(i, 2));
				}
				auto res = h.SearchBatch(s, g, sd);
				py::list out;
				for (size_t i = 0; i < n; i++)
					out.append(py::make_tuple(res[i].status, res[i].cost, res[i].n_expanded, res[i].n_path));
				return out;
			});

	// ---- grid A* (pyplanning.cpp:124-197): cost / heuristic are Python callables per edge, as in the reference ----
	py::class_<NullAction>(m, "NullAction").def(py::init<>());
	using AStarHeuristicN2 = AStarHeuristic<GridCellPosition>;
	struct AStarHeuristicN2Wrapper : AStarHeuristicN2 {
		using AStarHeuristicN2::AStarHeuristicN2;
		double GetHeuristicValue(const GridCellPosition& state) override { PYBIND11_OVERRIDE_PURE(double, AStarHeuristicN2, GetHeuristicValue, state); }
		void SetGoal(const GridCellPosition& goal) override { PYBIND11_OVERRIDE_PURE(void, AStarHeuristicN2, SetGoal, goal); }
	};
	py::class_<AStarHeuristicN2, Ref<AStarHeuristicN2>, AStarHeuristicN2Wrapper>(m, "AStarHeuristicN2").def(py::init<>());
	py::class_<AStarHeuristicFcnN2, Ref<AStarHeuristicFcnN2>, AStarHeuristicN2>(m, "AStarHeuristicFcnN2").def(py::init<CellCostFcn>());
	py::class_<AverageHeuristic<GridCellPosition>, Ref<AverageHeuristic<GridCellPosition>>, AStarHeuristicN2>(m, "AverageHeuristicN2");

	using AStarStatePropagatorN2 = AStarStatePropagator<GridCellPosition>;
	struct AStarStatePropagatorN2Wrapper : AStarStatePropagatorN2 {
		using AStarStatePropagatorN2::AStarStatePropagatorN2;
		using ReturnType = std::vector<std::tuple<GridCellPosition, NullAction, double>>;
		ReturnType GetNeighborStates(const GridCellPosition& cell) override { PYBIND11_OVERRIDE_PURE(ReturnType, AStarStatePropagatorN2, GetNeighborStates, cell); }
	};
	py::class_<AStarStatePropagatorN2, Ref<AStarStatePropagatorN2>, AStarStatePropagatorN2Wrapper>(m, "AStarStatePropagatorN2").def(py::init<>());
	py::class_<AStarStatePropagatorFcnN2, Ref<AStarStatePropagatorFcnN2>, AStarStatePropagatorN2>(m, "AStarStatePropagatorFcnN2")
		.def(py::init<const Ref<OccupancyMap>&, const CellCostFcn&>());

	struct PathPlannerN2BaseWrapper : PathPlannerN2Base {
		using PathPlannerN2Base::PathPlannerN2Base;
		Status SearchPath() override { PYBIND11_OVERRIDE_PURE(Status, PathPlannerN2Base, SearchPath); }
		std::vector<GridCellPosition> GetPath() const override { PYBIND11_OVERRIDE_PURE(std::vector<GridCellPosition>, PathPlannerN2Base, GetPath); }
	};
	py::class_<PathPlannerN2Base, PathPlannerN2BaseWrapper>(m, "PathPlannerN2Base")
		.def(py::init<>())
		.def("search_path", &PathPlannerN2Base::SearchPath)
		.def("get_path", &PathPlannerN2Base::GetPath)
		.def("set_init_state", &PathPlannerN2Base::SetInitState)
		.def("set_goal_state", &PathPlannerN2Base::SetGoalState);
	py::class_<AStarN2, PathPlannerN2Base>(m, "AStarN2")
		.def(py::init<>())
		.def("initialize", &AStarN2::Initialize)
		.def("get_explored_states", &AStarN2::GetExploredStates)
		.def("get_expansion_order", &AStarN2::GetExpansionOrder)
		.def("get_optimal_cost", &AStarN2::GetOptimalCost);
	py::class_<BidirectionalAStarN2, PathPlannerN2Base>(m, "BidirectionalAStarN2")
		.def(py::init<>())
		.def_static("get_average_heuristic_pair", &BidirectionalAStarN2::GetAverageHeuristicPair)
		.def("initialize", &BidirectionalAStarN2::Initialize)
		.def("get_explored_states", &BidirectionalAStarN2::GetExploredStates)
		.def("get_expansion_orders", &BidirectionalAStarN2::GetExpansionOrders)
		.def("get_optimal_cost", &BidirectionalAStarN2::GetOptimalCost);

	// beyond the reference's module: the same two searches for many (init, goal) pairs on the device (Euclidean cost / heuristic)
	py::class_<GridSearchResult>(m, "GridSearchResult")
		.def_readonly("status", &GridSearchResult::status)
		.def_readonly("cost", &GridSearchResult::cost)
		.def_readonly("path", &GridSearchResult::path)
		.def_readonly("expanded", &GridSearchResult::expanded)
		.def_readonly("expanded_reverse", &GridSearchResult::expandedReverse);
	py::class_<GridAStarBatchHip>(m, "GridAStarBatch")
		.def(py::init<const Ref<OccupancyMap>&>())
		.def("search_batch",
			[](GridAStarBatchHip& self, const std::vector<GridCellPosition>& inits, const std::vector<GridCellPosition>& goals, bool bidirectional, bool wantExpanded) {
				return self.SearchBatch(inits, goals, bidirectional, wantExpanded);
			},
			py::arg("inits"), py::arg("goals"), py::arg("bidirectional") = false, py::arg("want_expanded") = false);

	py::class_<RRTParameters>(m, "RRTParameters")
		.def(py::init<>())
		.def_readwrite("max_iteration", &RRTParameters::maxIteration)
		.def_readwrite("max_number_tree_node", &RRTParameters::maxNumberTreeNode)
		.def_readwrite("max_connection_distance", &RRTParameters::maxConnectionDistance)
		.def_readwrite("goal_bias", &RRTParameters::goalBias);
	py::class_<RRTStarParameters>(m, "RRTStarParameters")
		.def(py::init<>())
		.def_readwrite("max_iteration", &RRTStarParameters::maxIteration)
		.def_readwrite("max_number_tree_node", &RRTStarParameters::maxNumberTreeNode)
		.def_readwrite("max_connection_distance", &RRTStarParameters::maxConnectionDistance)
		.def_readwrite("goal_bias", &RRTStarParameters::goalBias)
		.def_readwrite("rewire", &RRTStarParameters::rewire)
		.def_readwrite("radius_gamma", &RRTStarParameters::radiusGamma);
	py::class_<RRTR2>(m, "RRTR2")
		.def(py::init<const Point2d&, const Point2d&, const Ref<StateValidatorOccupancyMap>&>(), py::arg("lower"), py::arg("upper"), py::arg("validator") = nullptr)
		.def("set_parameters", &RRTR2::SetParameters)
		.def("set_seed", &RRTR2::SetSeed)
		.def("set_init_state", &RRTR2::SetInitState)
		.def("set_goal_state", &RRTR2::SetGoalState)
		.def("search_path", &RRTR2::SearchPath)
		.def("get_path", &RRTR2::GetPath);
	py::class_<RRTStarR2>(m, "RRTStarR2")
		.def(py::init<const Point2d&, const Point2d&, const Ref<StateValidatorOccupancyMap>&>(), py::arg("lower"), py::arg("upper"), py::arg("validator") = nullptr)
		.def("set_parameters", &RRTStarR2::SetParameters)
		.def("set_seed", &RRTStarR2::SetSeed)
		.def("set_init_state", &RRTStarR2::SetInitState)
		.def("set_goal_state", &RRTStarR2::SetGoalState)
		.def("search_path", &RRTStarR2::SearchPath)
		.def("get_path", &RRTStarR2::GetPath);
}
