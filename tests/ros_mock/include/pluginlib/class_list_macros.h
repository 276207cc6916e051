// MOCK (tests/ros_mock), not ROS: the export macro only checks that the class derives from the base and can be built.
#pragma once
#define PLUGINLIB_EXPORT_CLASS(cls, base) namespace { inline base* rtuf_mock_plugin_factory() { return new cls(); } }
